for w in 1 2 3; do for pr in low normal; do echo "== workers $w priority $pr"; GCL_LOADER_WORKERS=$w GCL_LOADER_PRIORITY=$pr python3 tools/micro/e2e_probe.py 2>&1 | grep "^B\|^C"; done; done
