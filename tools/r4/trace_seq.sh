# kernel sequence of one graph replay of the headline step (names in launch order with durations)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/seq; mkdir -p gpurun_out/seq
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
fn = glob.glob('gpurun_out/seq/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
# last replay: take the last N kernels where N = kernels per step (find the last labels_kernel)
idx = [i for i, r in enumerate(rows) if 'labels' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
out = []
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('void hno::', '').replace('hno::', '')[:70]
    out.append('%8.1f us  dur %6.1f  gap %5.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    prev_end = e
open('gpurun_out/seq/sequence.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
print('kernels per step', b - a, 'step span us', (int(rows[b]['Start_Timestamp']) - t0) / 1e3)
PY
