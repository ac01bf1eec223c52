# kernel-level same-box A/B of two source trees: the working tree and the worktree _r4 (any commit, built): bash tools/r5/ab_trees.sh [pattern]
PAT=${1:-"plain run|dht_|spec_mid|sum of kernel"}
bash tools/r5/prof_step.sh r05_tree_a | grep -E "$PAT"
(cd _r4 && GRAFT_REPO_ROOT=$PWD bash tools/r5/prof_step.sh r05_tree_b | grep -E "$PAT")
rm -rf gpurun_out/r05_tree_b; cp -r _r4/gpurun_out/r05_tree_b gpurun_out/r05_tree_b 2>/dev/null
