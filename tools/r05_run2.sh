# round-5 measurement script (run on the GPU box through gpurun): GPU test suite on the new arithmetic, then same-box A/B
export GCL_PRECISION_LOG=$PWD/gpurun_out/r05_precision_errors_b.log
export GCL_FULL_BWD_PRECISIONS=fp16x3,f32,bf16x6
rm -f $GCL_PRECISION_LOG
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05_t1.log
tail -8 gpurun_out/r05_t1.log
B="bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
  (cd .ab_prev && python3 $B > ../gpurun_out/r05_b1_prev_$i.json 2> ../gpurun_out/r05_b1_prev_$i.err)
  python3 $B > gpurun_out/r05_b1_new_$i.json 2> gpurun_out/r05_b1_new_$i.err
  GCL_DW_ROWS=0 python3 $B > gpurun_out/r05_b1_rows0_$i.json 2> gpurun_out/r05_b1_rows0_$i.err
  GCL_AUX_CU_PCT=75 python3 $B > gpurun_out/r05_b1_cu75_$i.json 2> gpurun_out/r05_b1_cu75_$i.err
  GCL_AUX_CU_PCT=50 python3 $B > gpurun_out/r05_b1_cu50_$i.json 2> gpurun_out/r05_b1_cu50_$i.err
done
for f in gpurun_out/r05_b1_*.json; do echo $f; cut -c1-200 $f; done
grep -h "k_bwd_weight_rows" gpurun_out/r05_b1_new_1.err | head
grep -h " 96-> 64\| 64-> 32" gpurun_out/r05_b1_rows0_1.err | head
