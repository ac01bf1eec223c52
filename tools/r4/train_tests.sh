cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
HNO_TRAIN_GRAPH_DEBUG=1 python3 -m pytest tests/test_training_loop.py -m gpu -q -x 2>&1 | tail -30
