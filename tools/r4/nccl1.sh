cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
HNO_TRAIN_GRAPH_DEBUG=1 python3 -m pytest tests/test_hip_ops.py -m gpu -q -x -k "rccl_single_rank" 2>&1 | tail -30
