# same-box A/B of the eval loop: the tree of an earlier commit (.ab_prev, tools/ab_prev.sh export <commit>) vs the working tree
R=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2 3; do
  echo "== prev"; (cd $R/.ab_prev && python3 tools/micro/eval_tail_probe.py noprof 2>&1 | grep "pairs/s")
  echo "== new";  (cd $R && python3 tools/micro/eval_tail_probe.py noprof 2>&1 | grep "pairs/s")
done
