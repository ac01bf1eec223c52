cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/dpp gpurun_out/hdp; mkdir -p gpurun_out/dpp gpurun_out/hdp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dpp -- python3 bench.py --dp-path --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hdp -- python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
def load(d):
    fn = glob.glob(f'gpurun_out/{d}/**/*kernel_stats.csv', recursive=True)[0]
    return {r['Name'][:70]: (int(r['Calls']), float(r['TotalDurationNs']) / 1e3) for r in csv.DictReader(open(fn))}
a, b = load('dpp'), load('hdp')
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0))[1] + b.get(n, (0, 0))[1]))
print('%-72s %14s %14s' % ('kernel', 'dp calls/us', 'plain calls/us'))
for n in names[:40]:
    ca, ta = a.get(n, (0, 0)); cb, tb = b.get(n, (0, 0))
    if ca != cb or abs(ta - tb) > 0.05 * max(ta, tb, 1):
        print('%-72s %6d %8.0f %6d %8.0f' % (n, ca, ta, cb, tb))
print('total us', sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
PY
