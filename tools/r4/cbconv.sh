cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/bench_cb_conv.py $ARGS 2>&1 | grep flags
