# SQ counters of the eval loop's kernels (registration + inference):  gpurun -- "bash tools/r05_eval_pmc.sh"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_eval
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $GRAFT_REPO_ROOT/gpurun_out/pmc_eval -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/micro/eval_tail_probe.py noprof > $GRAFT_REPO_ROOT/gpurun_out/r05_eval_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/pmc_eval/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    nm = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).replace("gcl::", "")
    agg[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[nm] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
out = ["rocprofv3 --kernel-trace --pmc SQ_* -- python3 tools/micro/eval_tail_probe.py noprof (tools/r05_eval_pmc.sh); sums over all dispatches"]
for nm, c in rows[:24]:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    out.append(f"{nm[:44]:44s} dispatches={cnt[nm]:5d} wave_cycles={wc:.3e} wait_any={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} "
               f"active={c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} valu_insts={c.get('SQ_INSTS_VALU',0):.3e} salu_insts={c.get('SQ_INSTS_SALU',0):.3e} lds_conflict={c.get('SQ_LDS_BANK_CONFLICT',0):.2e}")
open("gpurun_out/r05_eval_pmc_sq.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/pmc_eval
