#!/bin/bash
A="--steps 30 --warmup 10 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
for kb in 0 4400 8800 17600; do
  if [ $kb = 0 ]; then export GCL_DW_RG128=0; else export GCL_DW_RG128=1; export GCL_DW_RG128_KB=$kb; fi
  python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print('RG128_KB=$kb', d['ms_per_step'], 'ms/step; aux wg128 ms', (r.get('overlapped_aux_stream_kernel_ms_last_step') or {}).get('k_conv_bwd_weight_wg128'))"
done
done
