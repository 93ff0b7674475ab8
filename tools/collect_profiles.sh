#!/bin/bash
# Collects what profiles/ is made from, on the GPU box:  gpurun -- 'bash tools/collect_profiles.sh g'
# then here:  python tools/summarize_profiles.py r06_<tag> 10   (steps + warm-up of the profiled command)
# Separate passes (gpurun refuses --pmc combined with other trace domains): kernel trace + stats, SQ counters, FETCH_SIZE,
# WRITE_SIZE.  The program itself follows "--" (no shell / env hop: the profiler's preload initialises the GPU first).
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A="--steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary"
python3 -c "import sys; sys.path.insert(0, '$R'); from gcl_amd import _lib; print(_lib.source_hash())" > $R/gpurun_out/csrc_sha16.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_final $R/gpurun_out/pmc_final1 $R/gpurun_out/pmc_final2 $R/gpurun_out/pmc_final3
python3 $R/bench.py > $R/gpurun_out/r06_bench_$TAG.json 2> $R/gpurun_out/r06_bench_$TAG.log
python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r06_bench_${TAG}_2.json 2>> $R/gpurun_out/r06_bench_$TAG.log
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o r --output-format csv -- python3 $R/bench.py $A > $R/gpurun_out/r06_prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_final1 -o r --output-format csv -- python3 $R/bench.py $A >> $R/gpurun_out/r06_prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_final2 -o r --output-format csv -- python3 $R/bench.py $A >> $R/gpurun_out/r06_prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_final3 -o r --output-format csv -- python3 $R/bench.py $A >> $R/gpurun_out/r06_prof_$TAG.log 2>&1
cd $R
# keep what travels back small: the per-dispatch traces are not needed, the stats / counter CSVs are
python3 tools/trace_gaps.py $(find gpurun_out/prof_final -name "*_kernel_trace.csv" | head -1) > gpurun_out/r06_gaps_$TAG.txt 2>&1
cat gpurun_out/r06_gaps_$TAG.txt
find gpurun_out/prof_final -name "*_kernel_trace.csv" -delete
du -sh gpurun_out/prof_final gpurun_out/pmc_final*
cut -c1-400 gpurun_out/r06_bench_$TAG.json
cut -c1-200 gpurun_out/r06_bench_${TAG}_2.json
