cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/mha; mkdir -p gpurun_out/mha
python3 tools/dbg/mha_pw_shapes.py 2>/dev/null | tail -20
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mha -- python3 tools/bench_models.py hartleymha > gpurun_out/mha/log.txt 2>&1
python3 - <<'PY'
import csv, glob
fn = glob.glob('gpurun_out/mha/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot / 1e6)
for r in rows[:32]:
    print('%-100s %6d calls %9.1f us total  avg %7.1f' % (r['Name'][:100], int(r['Calls']), float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3))
PY
grep -v Warn gpurun_out/mha/log.txt | tail -2 | cut -c1-300
