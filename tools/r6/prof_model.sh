# kernel stats of one model family's step: bash tools/r6/prof_model.sh <tag> <bench_models case>
TAG=$1; CASE=$2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o run -- python3 tools/bench_models.py $CASE > gpurun_out/$TAG/log.txt 2>&1
grep "^{" gpurun_out/$TAG/log.txt | cut -c1-200
python3 - <<PY
import csv, glob
f = sorted(glob.glob('gpurun_out/$TAG/prof/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print('%-100s %6d %9.1f us  %5.1f%%' % (r['Name'][:100], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
