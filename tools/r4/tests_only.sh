cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4t; mkdir -p $O
python3 -m pytest tests -m gpu -q ${PYTEST_ARGS} > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -30
