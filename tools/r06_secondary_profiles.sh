#!/bin/bash
# kernel stats of (1) the eval loop on twin pairs (configs[4]) and (2) the step fed from raw scans:
#   gpurun -- 'bash tools/r06_secondary_profiles.sh'  -> gpurun_out/r06_eval_twin_kernel_stats.txt, r06_e2e_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
summ() {   # $1 = profile dir, $2 = units (pairs / steps) the run made, $3 = output, $4 = title
python3 - <<PY
import csv, glob, re
ks = glob.glob("$1/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(ks)))
n = float("$2")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["$4", f"GPU kernel time per unit: {tot/1e6/n:.3f} ms"]
for r in rows[:34]:
    nm = re.sub(r"^void ", "", r["Name"].split("(")[0]).replace("gcl::", "").replace(" ", "")
    out.append(f"{nm[:60]:60s} calls/unit={int(r['Calls'])/n:7.2f} us/unit={float(r['TotalDurationNs'])/1e3/n:9.1f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
open("$3", "w").write("\n".join(out) + "\n")
print("\n".join(out[:16]))
PY
}
rm -rf $R/gpurun_out/prof_twin
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_twin -o r --output-format csv -- python3 $R/tools/micro/twin_eval_probe.py 0.5 > $R/gpurun_out/r06_twin_prof.log 2>&1
# twin_eval_probe at one share: 1 collect pass + (1 warm + 5 x 4) x 2 batch sizes = 43 loops of 8 pairs = 344 pairs
summ $R/gpurun_out/prof_twin 344 $R/gpurun_out/r06_eval_twin_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python3 tools/micro/twin_eval_probe.py 0.5 (eval_pairs on twin pairs, inlier share 0.16, batch_pairs 8 and 1; per pair over 344 pairs)"
rm -rf $R/gpurun_out/prof_e2e
E2E_ONLY=B rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_e2e -o r --output-format csv -- python3 $R/tools/micro/e2e_probe.py > $R/gpurun_out/r06_e2e_prof.log 2>&1
summ $R/gpurun_out/prof_e2e 84 $R/gpurun_out/r06_e2e_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python3 tools/micro/e2e_probe.py with E2E_ONLY=B (train_from_scans, 2 x (10 + 30 + 2) steps; per step over 84 steps; includes the 2 set-up builds)"
find $R/gpurun_out/prof_twin $R/gpurun_out/prof_e2e -name "*_kernel_trace.csv" -delete
