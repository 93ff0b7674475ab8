"""DIAGNOSTIC: per-step GPU busy time, idle gaps and top kernels from a rocprofv3 --kernel-trace sqlite DB.
usage: python tools/trace_gaps.py gpurun_out/prof_final/r_results.db [n_steps]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()


def short(n):
    n = re.sub(r"^void ", "", n.split("(")[0]).replace("gcl::", "")
    m = re.match(r"_ZN3gcl\d+([a-z_0-9]+?)(I[^E]*E)?E", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:50]


# the last step = from the last k_coords_insert to the end
starts = [i for i, r in enumerate(rows) if "k_coords_insert" in r[0]]
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
wall = step[-1][2] - step[0][1]
busy = sum(e - s for _, s, e in step)
print(f"last full step: {len(step)} kernels, wall {wall / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(wall - busy) / 1e6:.2f} ms")
gaps = []
for a, b in zip(step[:-1], step[1:]):
    g = b[1] - a[2]
    if g > 0:
        gaps.append((g, short(a[0]), short(b[0])))
gaps.sort(reverse=True)
print("largest gaps (us): ", [(round(g / 1e3, 1), a, b) for g, a, b in gaps[:12]])
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<2us" if g < 2e3 else "<5us" if g < 5e3 else "<10us" if g < 1e4 else "<50us" if g < 5e4 else ">=50us"] += g
print("idle by gap size (ms):", {k: round(v / 1e6, 2) for k, v in hist.items()})
tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in step:
    tot[short(n)][0] += 1
    tot[short(n)][1] += e - s
for n, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:50s} calls={v[0]:4d} ms={v[1] / 1e6:7.3f} avg_us={v[1] / v[0] / 1e3:8.1f}")
