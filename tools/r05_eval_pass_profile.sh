# kernels of ONE inference pass over EP_PAIRS pairs (default 8):  gpurun -- "bash tools/r05_eval_pass_profile.sh"
export EP_PAIRS=${EP_PAIRS:-8} EP_PASSES=40
python3 tools/micro/eval_pass.py
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_pass
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pass -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/micro/eval_pass.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_pass -name "*_kernel_trace.csv" -delete
python3 - <<'PY'
import csv,glob,re
f=glob.glob("gpurun_out/prof_pass/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
n=44
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per pass: {tot/1e6/n:.3f} ms")
for r in rows[:30]:
    nm=re.sub(r"^void ","",r["Name"].split("(")[0]).replace("gcl::","")
    print(f"{nm[:56]:56s} calls/pass={int(r['Calls'])/n:6.1f} us/pass={float(r['TotalDurationNs'])/1e3/n:8.1f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
rm -rf gpurun_out/prof_pass
