python -m pytest tests -m gpu -x -q -k "inference_plan or eval_pairs or oracle_chain" 2>&1 | tail -3
for i in 1 2; do python3 bench.py --secondary-worker 2>/dev/null | tail -1 | python3 -c "import sys, json; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in d if k.startswith('configs')}, indent=1))"; done
for s in 1 2 3; do echo "GCL_FWD_STREAMS=$s"; GCL_FWD_STREAMS=$s python3 tools/micro/eval_tail_probe.py noprof 2>&1 | grep pairs/s; done
