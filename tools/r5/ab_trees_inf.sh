# same-box A/B of two source trees (working tree, worktree _r4) on the headline step and on the inference protocol
bash tools/r5/ab_trees.sh "plain run|dht_inv|sum of kernel"
cd $GRAFT_REPO_ROOT
for t in . _r4 . _r4; do (cd $t && python3 tools/bench_infer.py --samples 16 | cut -c150-215); done
