# one bench_models case under environment settings, same box, interleaved: bash tools/r6/model_env_ab.sh <case> "A=1" "A=0" ...
CASE=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for kv in "$@"; do
  env $kv python3 tools/bench_models.py $CASE 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d.get('ms_per_step_graph'), d.get('ms_per_step'), d.get('loss'), d.get('error'))"
done; done
