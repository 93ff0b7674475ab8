# experiment 49: presplit threshold with producer-written planes (bound mode writes the planes of every BatchNorm output >= the
# threshold), alternating with the default and with the round's first commit (.ab_prev/, tools/ab_prev.sh export)
B="bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
  python3 $B > gpurun_out/r05_b3_c128_$i.json 2> gpurun_out/r05_b3_c128_$i.err
  GCL_PRESPLIT_MIN_C=64 python3 $B > gpurun_out/r05_b3_c64_$i.json 2> gpurun_out/r05_b3_c64_$i.err
  GCL_PRESPLIT_MIN_C=32 python3 $B > gpurun_out/r05_b3_c32_$i.json 2> gpurun_out/r05_b3_c32_$i.err
  (cd .ab_prev && python3 $B > ../gpurun_out/r05_b3_prev_$i.json 2> ../gpurun_out/r05_b3_prev_$i.err)
done
for f in gpurun_out/r05_b3_*.json; do echo $f; cut -c1-200 $f; done
