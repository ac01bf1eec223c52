cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_hip_ops.py -m gpu -q -k "attention or mha" 2>&1 | tail -4
python3 tools/dbg/hmha_one.py 2>&1 | tail -3
python3 tools/bench_models.py hartleymha 2>&1 | tail -1 | cut -c1-400
