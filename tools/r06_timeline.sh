#!/bin/bash
# per-step queue timeline of the training step:  gpurun -- 'bash tools/r06_timeline.sh'  -> gpurun_out/r06_step_timeline.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A="--steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events --no-secondary"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o r --output-format csv -- python3 $R/bench.py $A > $R/gpurun_out/r06_tl.log 2>&1
cd $R
python3 tools/step_timeline.py $(find gpurun_out/prof_tl -name "*_kernel_trace.csv" | head -1) > gpurun_out/r06_step_timeline.txt 2>&1
find gpurun_out/prof_tl -name "*_kernel_trace.csv" -delete
head -70 gpurun_out/r06_step_timeline.txt
