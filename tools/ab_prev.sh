#!/bin/bash
# Same-box A/B of the working tree against an earlier commit:  here:   bash tools/ab_prev.sh export <commit>
#                                                               there:  gpurun -- 'bash tools/ab_prev.sh run [rounds]'
# `export` unpacks <commit> into .ab_prev/ (git-ignored, travels with the snapshot) and builds its library there; `run`
# alternates `bench.py --no-cpu-baseline --no-secondary` of the two trees and prints the ms/step of every run.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
if [ "$1" = "export" ]; then
  rm -rf .ab_prev && mkdir .ab_prev && git archive ${2:-HEAD} | tar -x -C .ab_prev && rm -rf .ab_prev/tests/golden/full_bs4_* .ab_prev/profiles
  (cd .ab_prev && python3 -c "import sys; sys.path.insert(0, '.'); from gcl_amd import _lib; _lib.build(force=False, verbose=False)" && make -s -C oracle)
  exit $?
fi
B="bench.py --no-cpu-baseline --no-secondary"
for i in $(seq 1 ${2:-2}); do
  (cd .ab_prev && python3 $B > ../gpurun_out/ab_prev_$i.json 2> ../gpurun_out/ab_prev_$i.err)
  python3 $B > gpurun_out/ab_new_$i.json 2> gpurun_out/ab_new_$i.err
done
for f in gpurun_out/ab_prev_*.json gpurun_out/ab_new_*.json; do echo "$f $(python3 -c "import json,sys; print(json.load(open('$f'))['ms_per_step'])")"; done
