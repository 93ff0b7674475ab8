#!/bin/bash
# small launches with narrower column blocks (GCL_NB_SMALL_WGS, conv.hip conv_fwd_nb): one pair / eight pairs per inference pass
# and the training step, alternating on one box:  gpurun -- 'bash tools/r06_ab_small_nb.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for i in 1 2 3; do
  for v in 0 256 512; do
    echo "small_wgs=$v one pair:    $(GCL_NB_SMALL_WGS=$v python3 tools/micro/eval_pass.py 2>&1 | grep 'ms per pass')"
  done
done
for v in 0 256 512; do
  echo "small_wgs=$v eight pairs: $(GCL_NB_SMALL_WGS=$v EP_PAIRS=8 python3 tools/micro/eval_pass.py 2>&1 | grep 'ms per pass')"
done
for i in 1 2; do
  for v in 0 256 512; do
    echo "small_wgs=$v step: $(GCL_NB_SMALL_WGS=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  done
done
