cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); c=d['config']; print('$1', d['value'], d['ms_per_step'], '|', c['launch'], '| ranks', c['rccl_ranks'], '|', c['allreduce_in_graph_chosen_by'], '|', c['schedule'])"; }
python3 bench.py --dp-path --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | show dp_default
HNO_DP_CAPTURE_ALLREDUCE=1 python3 bench.py --dp-path --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | show dp_one_replay
