#!/bin/bash
# per-dispatch timeline of ONE inference pass over one pair (which queue, start, duration, gap to the previous dispatch on the
# same queue):  gpurun -- 'bash tools/r06_pass_timeline.sh'  -> gpurun_out/r06_pass_timeline.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export EP_PAIRS=${EP_PAIRS:-1} EP_PASSES=12
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_pass
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_pass -o r --output-format csv -- python3 $R/tools/micro/eval_pass.py > /dev/null 2>&1
cd $R
python3 - <<'PY' > gpurun_out/r06_pass_timeline.txt
import csv, glob, re
f = glob.glob("gpurun_out/prof_pass/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).replace("gcl::", "")[:44]
# a pass starts at every k_stem_fwd_occ's preceding k_coords_insert; simpler: cut at the first-layer kernel
stems = [i for i, r in enumerate(rows) if "k_stem_fwd" in r["Kernel_Name"]]
a, b = stems[-3], stems[-2]
# walk back from the stem launch to the first dispatch of its pass: the first k_coords_insert after the previous row-normalise
start = a
while start > 0 and "k_row_normalize" not in rows[start - 1]["Kernel_Name"]:
    start -= 1
end = b
while end > 0 and "k_row_normalize" not in rows[end - 1]["Kernel_Name"]:
    end -= 1
sel = rows[start:end]
t0 = int(sel[0]["Start_Timestamp"])
last = {}
print(f"one pass: {len(sel)} dispatches, {(int(sel[-1]['End_Timestamp']) - t0) / 1e3:.1f} us from first start to last end")
for r in sel:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    print(f"q{q:>2s} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {nm(r)}")
PY
find gpurun_out/prof_pass -name "*_kernel_trace.csv" -delete
head -5 gpurun_out/r06_pass_timeline.txt
