# round-5 final collection on ONE box: GPU suite, the driver's command, profiles (kernel stats + PMC passes), exclusive
# (single-stream) kernel stats, the eval tail's kernel stats
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-f}
export GCL_PRECISION_LOG=$R/gpurun_out/r05_precision_errors_final.log
rm -f $GCL_PRECISION_LOG
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r05_t_final.log
tail -6 gpurun_out/r05_t_final.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_final_driver_cmd.json 2> gpurun_out/r05_bench_final_driver_cmd.log
cut -c1-300 gpurun_out/r05_bench_final_driver_cmd.json
bash tools/collect_profiles.sh $TAG > gpurun_out/r05_collect_$TAG.log 2>&1
tail -12 gpurun_out/r05_collect_$TAG.log | cut -c1-300
export GCL_PLAN_AUX=0
bash tools/prof_stats.sh r05_noaux > gpurun_out/r05_noaux.log 2>&1
unset GCL_PLAN_AUX
head -30 gpurun_out/r05_noaux_stats.txt
bash tools/eval_tail_profile.sh > gpurun_out/r05_eval_final.txt 2>&1
grep -A28 "GPU busy" gpurun_out/r05_eval_final.txt | cut -c1-160
head -4 gpurun_out/r05_eval_probe.txt
