# kernel stats + host profile of the packaged eval loop (configs[4]) on the GPU box:  gpurun -- "bash tools/eval_tail_profile.sh"
python3 tools/micro/eval_tail_probe.py > gpurun_out/r05_eval_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_eval
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_eval -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/micro/eval_tail_probe.py noprof > $GRAFT_REPO_ROOT/gpurun_out/r05_eval_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_eval -name "*_kernel_trace.csv" -delete
head -60 gpurun_out/r05_eval_probe.txt
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_eval/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("GPU busy total ms", tot/1e6)
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
python -m pytest tests/test_gpu_plan.py -x -q -k "inference_plan" 2>&1 | tail -3
