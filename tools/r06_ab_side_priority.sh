A="--steps 40 --warmup 10 --no-cpu-baseline --no-secondary"
for rep in 1 2; do for pr in low high normal; do
  GCL_SIDE_PRIORITY=$pr python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GCL_SIDE_PRIORITY=$pr', d['ms_per_step'], 'ms/step')"
done; done
