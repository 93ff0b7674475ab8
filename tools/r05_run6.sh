python -m pytest tests -m gpu -x -q -k "inference_plan or sc2pcr or eval or extract_features or valid_epoch or forward_clouds" 2>&1 | tail -15
echo "---- bench secondary, sparse on"
python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r05_b4.json 2> gpurun_out/r05_b4.err; echo "rc=$?"; tail -3 gpurun_out/r05_b4.err | cut -c1-300
python3 -c "
import json; d=json.load(open('gpurun_out/r05_b4.json')); print(json.dumps(d.get('secondary'), indent=1))"
echo "---- bench secondary, GCL_SC2_SPARSE=0 GCL_SC2_REFINE_ONE_LAUNCH=0"
GCL_SC2_SPARSE=0 GCL_SC2_REFINE_ONE_LAUNCH=0 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r05_b4_old.json 2> gpurun_out/r05_b4_old.err; echo "rc=$?"; tail -3 gpurun_out/r05_b4_old.err | cut -c1-300
python3 -c "
import json; d=json.load(open('gpurun_out/r05_b4_old.json')); print(json.dumps(d.get('secondary'), indent=1))"
