python -m pytest tests -m gpu -x -q -k "sc2pcr or eval_pairs or oracle_chain or valid_epoch" 2>&1 | tail -5
bash tools/r05_run4.sh 2>&1 | tail -45
