#!/bin/bash
# same-box alternation: map helper threads 1 / 2 / 3, prefetch depth 2 / 3 (bench headline only)
A="--steps 40 --warmup 10 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
for cfg in "1 2" "2 2" "2 3" "3 3" "1 3"; do
  set -- $cfg
  GCL_MAP_WORKERS=$1 GCL_PREFETCH_DEPTH=$2 python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GCL_MAP_WORKERS=$1 GCL_PREFETCH_DEPTH=$2', d['ms_per_step'], 'ms/step', d['config']['host_enqueue_ms_per_step'])"
done
done
