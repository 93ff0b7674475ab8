#!/bin/bash
# kernel-trace + stats pass only:  gpurun -- 'bash tools/prof_stats.sh <tag>'  -> gpurun_out/<tag>_stats.txt
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A="--steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o r --output-format csv -- python3 $R/bench.py $A > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob, re
ks = glob.glob("gpurun_out/prof_$TAG/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(ks)))
n = 10
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"GPU busy per step: {tot/1e6/n:.2f} ms"]
for r in rows[:60]:
    nm = re.sub(r"^void ", "", r["Name"].split("(")[0]).replace("gcl::", "").replace(" ", "")
    out.append(f"{nm[:62]:62s} calls/step={int(r['Calls'])/n:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/n:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
open("gpurun_out/${TAG}_stats.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:45]))
PY
find gpurun_out/prof_$TAG -name "*_kernel_trace.csv" -delete
