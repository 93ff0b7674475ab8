B="bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
  python3 $B > gpurun_out/r05_b6_default_$i.json 2> /dev/null
  GCL_PLAN_AUX=low python3 $B > gpurun_out/r05_b6_low_$i.json 2> /dev/null
  GCL_PLAN_AUX=high python3 $B > gpurun_out/r05_b6_high_$i.json 2> /dev/null
done
for f in gpurun_out/r05_b6_*.json; do echo "$f $(python3 -c "import json; print(json.load(open('$f'))['ms_per_step'])")"; done
