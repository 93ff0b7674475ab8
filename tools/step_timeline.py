"""DIAGNOSTIC: per-step timeline of a rocprofv3 kernel trace of bench.py -- which queue ends a step, how long each queue is
busy inside it, and the idle stretches of the main queue.
usage: python tools/step_timeline.py <..._kernel_trace.csv> [steps_from_the_end]

A step is cut at the start of k_sgd_multi (one per optimizer step, main queue).  For the last steps of the trace it prints, per
hardware queue: dispatches, busy time, first start and last end relative to the step's start, and the last kernel; then every
idle stretch of the main queue longer than 20 us with the kernels either side, and the time for which ONLY a non-main queue
was running (the tail the main stream waits for at a join)."""
import csv
import sys


def main():
    path = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
    ev.sort()
    sgd = [e for e in ev if "k_sgd_multi" in e[3]]
    if len(sgd) < 2:
        print("fewer than two optimizer steps in the trace")
        return
    qmain = sgd[0][2]
    for i in range(max(1, len(sgd) - last), len(sgd)):
        s0, s1 = sgd[i - 1][0], sgd[i][0]
        step = [e for e in ev if s0 <= e[0] < s1]
        print(f"step {i}: {(s1 - s0) / 1e6:.3f} ms between optimizer launches, {len(step)} dispatches")
        for q in sorted(set(e[2] for e in step)):
            k = [e for e in step if e[2] == q]
            busy = sum(e[1] - e[0] for e in k)
            le = max(k, key=lambda e: e[1])
            print(f"  queue {q}{' (main)' if q == qmain else ''}: n = {len(k)}, busy {busy / 1e6:.3f} ms, first start +"
                  f"{(k[0][0] - s0) / 1e6:.3f}, last end +{(le[1] - s0) / 1e6:.3f} ({le[3][:60]})")
        k = [e for e in step if e[2] == qmain]
        others = [e for e in step if e[2] != qmain]
        idle_total, covered = 0, 0
        for j in range(len(k) - 1):
            g0, g1 = k[j][1], k[j + 1][0]
            if g1 - g0 <= 0:
                continue
            idle_total += g1 - g0
            # part of the gap during which some other queue runs a kernel
            iv = sorted((max(a, g0), min(b, g1)) for a, b, _, _ in others if b > g0 and a < g1)
            cov, cur = 0, g0
            for a, b in iv:
                a = max(a, cur)
                if b > a:
                    cov += b - a
                    cur = b
            covered += cov
            if g1 - g0 > 20000:
                print(f"    main idle {(g1 - g0) / 1e3:6.0f} us at +{(g0 - s0) / 1e6:.3f} ({cov / 1e3:.0f} us of it with another queue "
                      f"running): after {k[j][3][:44]} | before {k[j + 1][3][:44]}")
        print(f"  main queue idle inside the step: {idle_total / 1e6:.3f} ms, of which {covered / 1e6:.3f} ms while another queue was "
              f"running a kernel")


if __name__ == "__main__":
    main()
