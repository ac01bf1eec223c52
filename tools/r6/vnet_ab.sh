# V-Net-DS cfg4 bf16: A/B of environment settings on the graph-replayed step + the deep layers alone: bash tools/r6/vnet_ab.sh "A=1" "A=0"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  echo "== $kv"; env $kv python3 tools/bench_cb_conv.py l2_96_96 l3_192_192 l4_384_384 2>/dev/null
done
for rep in 1 2; do
for kv in "$@"; do
  env $kv python3 tools/bench_models.py vnetds_cfg4:bf16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d.get('ms_per_step_graph'), d.get('ms_per_step'), d.get('error'))"
done; done
