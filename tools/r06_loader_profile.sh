#!/bin/bash
# loader timing + kernel stats:  gpurun -- 'bash tools/r06_loader_profile.sh'  -> gpurun_out/r06_loader_*.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $R/tools/loader_bench.py > $R/gpurun_out/r06_loader_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_loader
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_loader -o r --output-format csv -- python3 $R/tools/loader_bench.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, re
ks = glob.glob("$R/gpurun_out/prof_loader/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(ks)))
out = ["rocprofv3 --kernel-trace --stats -- python3 tools/loader_bench.py (12 per-sample builds x 2 forms, 12 batch builds of 4 samples)"]
for r in rows[:30]:
    nm = re.sub(r"^void ", "", r["Name"].split("(")[0]).replace("gcl::", "").replace(" ", "")
    out.append(f"{nm[:58]:58s} calls={int(r['Calls']):6d} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
open("$R/gpurun_out/r06_loader_kernel_stats.txt", "w").write("\n".join(out) + "\n")
PY
find $R/gpurun_out/prof_loader -name "*_kernel_trace.csv" -delete
cat $R/gpurun_out/r06_loader_bench.txt $R/gpurun_out/r06_loader_kernel_stats.txt
