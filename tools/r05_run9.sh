python -m pytest tests -m gpu -x -q -k "sc2pcr or eval_pairs or oracle_chain or valid_epoch or inference_plan" 2>&1 | tail -5
bash tools/r05_run4.sh 2>&1 | grep -v "^ \|^$" | head -40
for s in 1 2 3 4; do echo "GCL_EVAL_STREAMS=$s"; GCL_EVAL_STREAMS=$s python3 tools/micro/eval_tail_probe.py noprof 2>&1 | grep pairs/s; done
python3 bench.py --secondary-worker 2>/dev/null | tail -1 | python3 -c "import sys, json; print(json.dumps(json.loads(sys.stdin.read()), indent=1))"
