cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); print('$1', d['ms_per_step'], [v for k,v in d['secondary'].items() if 'ms_per_step' in k and 'cfg' in k])"; }
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | show default
HNO_BENCH_SEC_NOALGO=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | show noalgo
