cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_hip_ops.py tests/test_bf16_models.py -x -q -m gpu -k "pwconv or hnosegxs or dma_ring or channel_padded or chain or branch or block" 2>&1 | tail -2
bash tools/r5/ab_trees.sh 2>&1 | tail -12
