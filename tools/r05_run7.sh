export GCL_PRECISION_LOG=$PWD/gpurun_out/r05_precision_errors_d.log
rm -f $GCL_PRECISION_LOG
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05_t3.log
tail -8 gpurun_out/r05_t3.log
B="bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
  (cd .ab_prev && python3 $B > ../gpurun_out/r05_b5_prev_$i.json 2> ../gpurun_out/r05_b5_prev_$i.err)
  python3 $B > gpurun_out/r05_b5_new_$i.json 2> gpurun_out/r05_b5_new_$i.err
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_b5_driver.json 2> gpurun_out/r05_b5_driver.err
for f in gpurun_out/r05_b5_*.json; do echo $f; cut -c1-200 $f; done
python3 -c "
import json; d=json.load(open('gpurun_out/r05_b5_driver.json')); print(json.dumps(d.get('secondary'), indent=1)); print(d.get('cpu_baseline')); print({k: d['roofline'][k] for k in ('kernel','frac','traffic','mfma_busy','avg_launch_us')})"
