"""DIAGNOSTIC: the main queue's dispatches between the forward row normalisation and its backward (the loss section of a
training step) in a rocprofv3 kernel trace of bench.py: name, duration, gap to the previous dispatch.
usage: python tools/loss_section.py <..._kernel_trace.csv>"""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
ev.sort()
sgd = [e for e in ev if "k_sgd_multi" in e[3]]
q = sgd[0][2]
main = [e for e in ev if e[2] == q and e[0] > sgd[-2][0] and e[0] < sgd[-1][0]]
i0 = next(i for i, e in enumerate(main) if "k_row_normalize<false>" in e[3])
i1 = next(i for i, e in enumerate(main) if "k_row_normalize<true>" in e[3])
print(f"loss section: {(main[i1][0] - main[i0][1]) / 1e3:.1f} us between the two row-normalise kernels, {i1 - i0 - 1} dispatches")
busy = 0
for j in range(i0, i1 + 1):
    e = main[j]
    gap = (e[0] - main[j - 1][1]) / 1e3
    busy += (e[1] - e[0]) if i0 < j < i1 else 0
    print(f"  gap {gap:6.1f} us  run {(e[1] - e[0]) / 1e3:6.1f} us  {e[3][:90]}")
print(f"kernel time inside: {busy / 1e3:.1f} us")
