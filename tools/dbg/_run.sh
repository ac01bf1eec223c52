cd /tmp && export TMPDIR=/tmp
cd /root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp -o dp -- python3 bench.py --dp-path --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-secondary > /tmp/dp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1 -o one -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-secondary > /tmp/one.log 2>&1
python3 - <<'PY'
import csv, glob
def load(d):
    out = {}
    for fn in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            out[r['Name'][:70]] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e3)
    return out
a, b = load('/tmp/prof_dp'), load('/tmp/prof_1')
print('total us: dp', sum(v[1] for v in a.values()), 'one', sum(v[1] for v in b.values()))
for k in sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))[:14]:
    print(f'{k:70s} dp {a.get(k, (0, 0))}  one {b.get(k, (0, 0))}')
PY
tail -1 /tmp/dp.log | cut -c1-200; tail -1 /tmp/one.log | cut -c1-200
