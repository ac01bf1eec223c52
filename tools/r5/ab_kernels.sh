# kernel-level same-box A/B of two environment settings: bash tools/r5/ab_kernels.sh "A=1" "A=0" [grep pattern]
PAT=${3:-"plain run|dht_|spec_mid|sum of kernel"}
bash tools/r5/prof_step.sh r05_ab_a "$1" | grep -E "$PAT"
bash tools/r5/prof_step.sh r05_ab_b "$2" | grep -E "$PAT"
