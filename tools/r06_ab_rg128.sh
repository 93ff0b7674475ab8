#!/bin/bash
# same-box alternation: range-grouped 128 x 128 weight gradient on / off (GCL_DW_RG128), bench headline only
A="--steps 30 --warmup 10 --no-cpu-baseline --no-secondary"
for i in 1 2; do
  for v in 1 0; do
    GCL_DW_RG128=$v python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print('GCL_DW_RG128=$v', d['ms_per_step'], 'ms/step; aux wg128 ms', (r.get('overlapped_aux_stream_kernel_ms_last_step') or {}).get('k_conv_bwd_weight_wg128'))"
  done
done
