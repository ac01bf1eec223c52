# same-box A/B of environment settings on the headline step: bash tools/r6/ab_env.sh "A=1" "A=0" ...   (SPLIT=1: two-stream schedule)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for kv in "$@"; do
  env $kv HNO_SPLIT_STREAMS=${SPLIT:-0} python3 bench.py --steps 30 --warmup 5 --bursts 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['value'], d['ms_per_step'], d['config']['ms_per_step_bursts'])"
done; done
