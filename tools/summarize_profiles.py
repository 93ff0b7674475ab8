"""Turns the rocprofv3 CSVs merged into gpurun_out/ into the small summaries committed under profiles/.

  gpurun_out/prof_final/*/…_kernel_stats.csv           (rocprofv3 --kernel-trace --stats -- python3 bench.py …)
  gpurun_out/pmc_final{1,2,3}/*/…_counter_collection.csv (separate --pmc passes: SQ_*, FETCH_SIZE, WRITE_SIZE)
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
n_steps_arg = int(sys.argv[2]) if len(sys.argv) > 2 else 5          # steps + warm-up of the profiled command
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def short(name):
    n = name.split("(")[0]
    n = re.sub(r"^void ", "", n).replace("gcl::", "").replace(" ", "")
    # k_conv_fwd_dma<NB, PRE, EPI, GRP>: the training launches are GRP = false; bench.py names them by the first three
    return re.sub(r"^(k_conv_fwd_dma<[^,]+,[^,]+,[^,]+),false>$", r"\1>", n)


ks = (glob.glob(os.path.join(G, "prof_final", "*", "*_kernel_stats.csv")) + glob.glob(os.path.join(G, "prof_final", "*_kernel_stats.csv")))[0]
shutil.copy(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(ks)))
n_steps = n_steps_arg
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(P, f"{tag}_kernel_stats_summary.txt"), "w") as fh:
    fh.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events "
             "(tools/collect_profiles.sh)\n")
    fh.write(f"GPU busy per training step: {tot / 1e6 / n_steps:.2f} ms ({n_steps} steps incl. warm-up)\n\n")
    for r in rows[:40]:
        fh.write(f"{short(r['Name'])[:60]:60s} calls/step={int(r['Calls']) / n_steps:7.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / n_steps:7.3f} "
                 f"avg_us={float(r['AverageNs']) / 1e3:8.1f} share={float(r['TotalDurationNs']) / tot * 100:5.1f}%\n")


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    hits = glob.glob(path) + glob.glob(path.replace(os.sep + "*" + os.sep, os.sep))
    for r in csv.DictReader(open(hits[0])):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


sq = load(os.path.join(G, "pmc_final1", "*", "*_counter_collection.csv"))
with open(os.path.join(P, f"{tag}_pmc_sq.txt"), "w") as fh:
    fh.write("rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU "
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -- python3 bench.py --steps 8 --warmup 2 (sums over all dispatches;\n"
             "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles)\n\n")
    for n, c in sorted(sq.items(), key=lambda kv: -sum(kv[1]["SQ_WAVE_CYCLES"])):
        wc = sum(c["SQ_WAVE_CYCLES"])
        if wc < 1e7:
            continue
        fh.write(f"{n[:52]:52s} dispatches={len(c['SQ_WAVE_CYCLES']):4d} wave_cycles={wc:.3e} wait_any={sum(c['SQ_WAIT_ANY']) / wc:.2f} "
                 f"wait_inst={sum(c['SQ_WAIT_INST_ANY']) / wc:.2f} active={sum(c['SQ_ACTIVE_INST_ANY']) / wc:.2f} "
                 f"mfma_busy_cycles={sum(c['SQ_VALU_MFMA_BUSY_CYCLES']):.3e} valu_insts={sum(c['SQ_INSTS_VALU']):.3e} "
                 f"lds_bank_conflict={sum(c['SQ_LDS_BANK_CONFLICT']):.3e}\n")
fe = load(os.path.join(G, "pmc_final2", "*", "*_counter_collection.csv"))
wr = load(os.path.join(G, "pmc_final3", "*", "*_counter_collection.csv"))
out = {"_source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `bench.py --steps 8 --warmup 2` (tools/collect_profiles.sh); "
                  "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 averaged over launches (gfx950: FETCH_SIZE "
                  "counts half of a 16-B/lane stream -- MI355X_MICROARCH.md 'HBM'; Infinity-Cache hits are included)"}
for n in fe:
    if True:      # every kernel of the profiled command (round 6: bench.py sums them into step_traffic_GB), torch's and the runtime's too
        f = sum(fe[n]["FETCH_SIZE"]) / len(fe[n]["FETCH_SIZE"])
        w = sum(wr[n]["WRITE_SIZE"]) / len(wr[n]["WRITE_SIZE"]) if n in wr else 0.0
        out[n] = {"launches": len(fe[n]["FETCH_SIZE"]), "fetch_size_kb_avg": round(f, 1), "write_size_kb_avg": round(w, 1),
                  "hbm_bytes_per_launch": round((2 * f + w) * 1024)}
        avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in rows}.get(n)
        if n in sq and avg_ns and sum(sq[n]["SQ_VALU_MFMA_BUSY_CYCLES"]) > 0:
            # MFMA pipe busy share: busy cycles per dispatch and SIMD (1024 SIMDs) over the launch duration at 2.4 GHz
            per = sum(sq[n]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(sq[n]["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0
            out[n]["mfma_busy"] = round(per / (avg_ns * 2.4), 4)
# stamp: the kernel sources the counters were collected on (tools/collect_profiles.sh writes it ON the GPU box, from the
# snapshot that ran); bench.py quotes traffic / mfma_busy only when its own sources have the same hash
stamp = os.path.join(G, "csrc_sha16.txt")
sys.path.insert(0, ROOT)
from gcl_amd import _lib  # noqa: E402
out["_csrc_sha16"] = open(stamp).read().strip() if os.path.exists(stamp) else _lib.source_hash()
out["_tag"] = tag
out["_steps"] = n_steps_arg
json.dump(out, open(os.path.join(P, "pmc_summary.json"), "w"), indent=1)
print(open(os.path.join(P, f"{tag}_kernel_stats_summary.txt")).read())
print(open(os.path.join(P, f"{tag}_pmc_sq.txt")).read()[:2500])
