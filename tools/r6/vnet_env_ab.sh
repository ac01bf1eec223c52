# V-Net-DS cfg4 bf16 graph-replayed step under environment settings, same box, interleaved: bash tools/r6/vnet_env_ab.sh "A=1" "A=0" ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for kv in "$@"; do
  env $kv python3 tools/bench_models.py vnetds_cfg4:bf16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d.get('ms_per_step_graph'), d.get('ms_per_step'), d.get('loss'), d.get('error'))"
done; done
timeout 900 python3 -m pytest tests/test_bf16_models.py tests/test_bf16_ops.py -x -q -m gpu 2>&1 | tail -3
