#!/bin/bash
# kernel stats of Matcher.SC2_PCR (11 registrations of 8000 correspondences) per inlier share:
#   gpurun -- 'bash tools/r06_sc2pcr_inlier_profile.sh [tag]'  -> gpurun_out/<tag>_sc2pcr_inlier_<share>.txt
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for S in 0.05 0.3 0.6; do
  rm -rf $R/gpurun_out/prof_sc2_$S
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc2_$S -o r --output-format csv -- python3 $R/tools/sc2pcr_profile.py $S > $R/gpurun_out/prof_sc2_$S.log 2>&1
  python3 - <<PY
import csv, glob, re
ks = glob.glob("$R/gpurun_out/prof_sc2_$S/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(ks)))
n = 11
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["rocprofv3 --kernel-trace --stats -- python3 tools/sc2pcr_profile.py $S  (11 registrations, 8000 correspondences, inlier share $S)",
       open("$R/gpurun_out/prof_sc2_$S.log").read().strip().splitlines()[0] if True else "",
       f"GPU busy per registration: {tot/1e3/n:.0f} us"]
for r in rows[:24]:
    nm = re.sub(r"^void ", "", r["Name"].split("(")[0]).replace("gcl::", "").replace(" ", "")
    out.append(f"{nm[:58]:58s} calls/reg={int(r['Calls'])/n:6.1f} us/reg={float(r['TotalDurationNs'])/1e3/n:8.1f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
open("$R/gpurun_out/${TAG}_sc2pcr_inlier_$S.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
  find $R/gpurun_out/prof_sc2_$S -name "*_kernel_trace.csv" -delete
done
