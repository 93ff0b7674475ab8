for s in 1 2 3 4 1 2; do echo "== GCL_EVAL_STREAMS=$s"; GCL_EVAL_STREAMS=$s python3 tools/micro/eval_tail_probe.py noprof 2>&1 | grep "pairs/s"; done
