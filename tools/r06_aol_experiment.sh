#!/bin/bash
# experiment 71:  gpurun -- 'bash tools/r06_aol_experiment.sh'  -> gpurun_out/r06_aol_experiment.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $R/tools/micro/aol_experiment.py > $R/gpurun_out/r06_aol_experiment.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_aol
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_aol -o r --output-format csv -- python3 $R/tools/micro/aol_experiment.py > /dev/null 2>&1
python3 - <<PY >> $R/gpurun_out/r06_aol_experiment.txt
import csv, glob, re
ks = glob.glob("$R/gpurun_out/prof_aol/**/*_kernel_stats.csv", recursive=True)[0]
print("rocprofv3 --kernel-trace --stats -- python3 tools/micro/aol_experiment.py (kernel durations over all three shapes, 11 launches each)")
for r in csv.DictReader(open(ks)):
    nm = re.sub(r"^void ", "", r["Name"].split("(")[0]).replace("gcl::", "").replace(" ", "")
    if "k_conv_fwd_dma" in nm or "elementwise" in nm or "k_amax" in nm:
        print(f"{nm[:70]:70s} calls={int(r['Calls']):5d} total_ms={float(r['TotalDurationNs'])/1e6:8.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
find $R/gpurun_out/prof_aol -name "*_kernel_trace.csv" -delete
cat $R/gpurun_out/r06_aol_experiment.txt
