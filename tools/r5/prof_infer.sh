# kernel stats of the inference protocol at the published size: bash tools/r5/prof_infer.sh <tag> [size...]
TAG=${1:-r05_inf}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
python3 tools/bench_infer.py --samples 8 "$@" > gpurun_out/$TAG/infer.jsonl 2>gpurun_out/$TAG/infer.err; cat gpurun_out/$TAG/infer.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o run -- python3 tools/bench_infer.py --samples 8 "$@" > gpurun_out/$TAG/prof.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob('gpurun_out/$TAG/prof/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel us per image', tot / 9 / 1e3)
for r in rows[:22]:
    print('%-90s %5d %9.1f us  %5.1f%%' % (r['Name'][:90], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
