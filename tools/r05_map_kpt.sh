# kernel-map lookups with 1 / 2 / 4 offsets per thread (GCL_MAP_KPT): kernel time of the map build alone (idle GPU)
#   gpurun -- "bash tools/r05_map_kpt.sh"
cd /tmp && export TMPDIR=/tmp
for k in ${KPTS:-1 2 4}; do
  export GCL_MAP_KPT=$k
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_map
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_map -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/micro/maps_helper_profile.py > /dev/null 2>&1
  echo "== GCL_MAP_KPT=$k"
  python3 - <<PY
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_map/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels, ms per map build: %.3f" % (tot/1e6/42))
for r in rows:
    if "k_kernel_map" in r["Name"] or "k_count_reduce" in r["Name"] or "k_permute" in r["Name"]:
        print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} ms_per_build={float(r['TotalDurationNs'])/1e6/42:7.3f} avg_us={float(r['AverageNs'])/1e3:7.1f}")
PY
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_map
