# same-box A/B of environment settings on the headline step (one pass, graph replay): bash tools/r5/ab_env.sh "A=1" "A=0" ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for kv in "$@"; do
  env $kv HNO_SPLIT_STREAMS=${SPLIT:-0} python3 bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['value'], d['ms_per_step'])"
done; done
