python3 tools/micro/twin_eval_probe.py > gpurun_out/r06_twin_probe.txt 2>&1
for s in 0.0 0.05 0.3 0.6; do python3 tools/sc2pcr_profile.py $s 2>&1 | grep "inlier share" >> gpurun_out/r06_twin_probe.txt; done
cat gpurun_out/r06_twin_probe.txt
