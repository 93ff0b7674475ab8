"""DIAGNOSTIC: every kernel dispatch and memory copy around the optimizer launch of the last steps of a trace
(rocprofv3 --kernel-trace --memory-copy-trace): what the GPU does between two training steps.
usage: python tools/boundary_ops.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [steps_from_the_end]"""
import csv, glob, sys

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ops = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"q{r['Queue_Id']}", r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    if rows:
        print("memory copy columns:", list(rows[0].keys()))
    for r in rows:
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy",
                    f"{r.get('Direction', '')} {r.get('Bytes', r.get('Size', ''))} B"))
ops.sort()
sgd = [o for o in ops if "k_sgd_multi" in o[3]]
for s in sgd[-last:]:
    t0 = s[0]
    print(f"--- optimizer launch at {t0}")
    for o in ops:
        if t0 - 400000 <= o[0] <= t0 + 1500000:
            print(f"  {(o[0] - t0) / 1e3:9.1f} us  +{(o[1] - o[0]) / 1e3:8.1f} us  {o[2]:5s} {o[3]}")
