"""GPU idle time between consecutive kernels, from a rocprofv3 --kernel-trace CSV (per queue and overall).
usage: python tools/trace_gaps.py <..._kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]                      # steady state only
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
span = t1 - t0
by_q = defaultdict(list)
for r in rows:
    by_q[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print(f"{len(rows)} dispatches over {span / 1e6:.2f} ms")
for q, iv in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e in iv)
    gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 50_000]
    print(f"queue {q}: {len(iv)} dispatches, busy {busy / 1e6:.2f} ms ({busy / span * 100:.1f} % of the span), "
          f"gaps < 50 us: {len(small)} totalling {sum(small) / 1e6:.2f} ms (median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.1f} us), "
          f"gaps >= 50 us: {len(pos) - len(small)} totalling {(sum(pos) - sum(small)) / 1e6:.2f} ms")
# union over all queues: time during which NO kernel runs
ev = sorted((s, e) for iv in by_q.values() for s, e in iv)
idle, cur = 0, ev[0][1]
for s, e in ev[1:]:
    if s > cur:
        idle += s - cur
    cur = max(cur, e)
print(f"no kernel running on any queue: {idle / 1e6:.2f} ms = {idle / span * 100:.1f} % of the span")
