# per-kernel durations of the feature 1-NN at the eval loop's shape:  gpurun -- "bash tools/nn_profile.sh"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_nn
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_nn -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/micro/nn_time.py 5000x5000x32 > $GRAFT_REPO_ROOT/gpurun_out/r05_nn_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_nn -name "*_kernel_trace.csv" -delete
tail -2 gpurun_out/r05_nn_prof.log
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_nn/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
