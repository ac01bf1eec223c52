# one-stream graph-replay kernel stats of the headline step: bash tools/r5/prof_step.sh <tag> [env assignments]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r05_p}; shift
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
env "$@" HNO_SPLIT_STREAMS=0 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain run:', d['value'], d['ms_per_step'])"
export HNO_SPLIT_STREAMS=0
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python3 - $O <<'PY'
import csv, glob, sys
fn = glob.glob(sys.argv[1] + '/graph/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
steps = [int(r['Calls']) for r in rows if 'labels4' in r['Name']][0]
tot = 0
for r in rows[:26]:
    per = int(r['Calls']) / steps
    print(r['Name'][:78].ljust(78), f"{per:6.2f}/step", f"{float(r['AverageNs'])/1e3:8.1f} us", f"{per*float(r['AverageNs'])/1e3:8.1f} us/step")
print('steps', steps, 'sum of kernel time per step (us):', sum(int(r['Calls']) * float(r['AverageNs']) for r in rows) / steps / 1e3)
PY
