# RECORD of a removed experiment (LESSONS 94): V-Net-DS cfg4 bf16 with the weight gradients on a side stream vs in the chain; the HNO_WGRAD_STREAM switch left the code with it
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for kv in HNO_WGRAD_STREAM=0 HNO_WGRAD_STREAM=1; do
  env $kv python3 tools/bench_models.py vnetds_cfg4:bf16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d.get('ms_per_step_graph'), d.get('ms_per_step'), d.get('loss'), d.get('max_mem_GB'), d.get('error'))"
done; done
timeout 900 python3 -m pytest tests/test_bf16_models.py tests/test_bf16_ops.py -x -q -m gpu 2>&1 | tail -3
