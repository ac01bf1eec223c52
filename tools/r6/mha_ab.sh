cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "junctions or mha or noseg_models or models_2d" 2>&1 | tail -8
for rep in 1 2; do
for j in 1 0; do
  HNO_JUNCTIONS=$j python3 tools/bench_models.py hartleymha 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('JUNCTIONS=$j', d.get('ms_per_step_graph'), d.get('ms_per_step'), d.get('error'))"
done; done
python3 tools/r6/mha_census.py 2>&1 | grep "^[0-9]" | head -12
