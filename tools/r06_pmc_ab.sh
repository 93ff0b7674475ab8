#!/bin/bash
# counter traffic of selected kernels under an environment switch:  gpurun -- 'bash tools/r06_pmc_ab.sh GCL_DW_RG128 "1 0" wg128'
# FETCH_SIZE and WRITE_SIZE in separate passes (bench.py --steps 4 --warmup 1); traffic = (2 FETCH + WRITE) KB per launch
VAR=$1; VALS=$2; PAT=$3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A="--steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary"
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  export $VAR=$v
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcab_${v}_$c
    rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcab_${v}_$c -o r --output-format csv -- python3 $R/bench.py $A > /dev/null 2>&1
  done
  python3 - <<PY
import csv, glob, collections, re
def load(c):
    f = glob.glob("$R/gpurun_out/pmcab_${v}_%s/**/*_counter_collection.csv" % c, recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).replace("gcl::", "").replace(" ", "")].append(float(r["Counter_Value"]))
    return d
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
tot = 0.0
for n in sorted(fe):
    f = sum(fe[n]) / len(fe[n]); w = sum(wr.get(n, [0])) / max(1, len(wr.get(n, [0])))
    tot += (2 * sum(fe[n]) + sum(wr.get(n, [0]))) * 1024 / 5
    if re.search("$PAT", n):
        print(f"$VAR=$v {n[:50]:50s} launches={len(fe[n]):4d} traffic/launch={(2*f+w)*1024/1e6:8.1f} MB (read {2*f*1024/1e6:8.1f} write {w*1024/1e6:7.1f})")
print(f"$VAR=$v all kernels: {tot/1e9:.2f} GB per step")
PY
  find $R/gpurun_out/pmcab_${v}_* -name "*.csv" -delete
done
