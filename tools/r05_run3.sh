# round-5 measurement script: GPU suite, then same-box A/B of the BatchNorm bound mode (plane images from the producer)
export GCL_PRECISION_LOG=$PWD/gpurun_out/r05_precision_errors_c.log
rm -f $GCL_PRECISION_LOG
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05_t2.log
tail -8 gpurun_out/r05_t2.log
B="bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
  (cd .ab_prev && python3 $B > ../gpurun_out/r05_b2_prev_$i.json 2> ../gpurun_out/r05_b2_prev_$i.err)
  python3 $B > gpurun_out/r05_b2_new_$i.json 2> gpurun_out/r05_b2_new_$i.err
  GCL_BN_PLANES=0 python3 $B > gpurun_out/r05_b2_noplanes_$i.json 2> gpurun_out/r05_b2_noplanes_$i.err
done
python3 bench.py --no-cpu-baseline > gpurun_out/r05_b2_secondary.json 2> gpurun_out/r05_b2_secondary.err
for f in gpurun_out/r05_b2_*.json; do echo $f; cut -c1-200 $f; done
python3 -c "
import json; d=json.load(open('gpurun_out/r05_b2_secondary.json')); print(json.dumps(d.get('secondary'), indent=1))"
