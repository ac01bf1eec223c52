# kernel sequence of one 240 x 240 x 155 inference pass (launch order, durations)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/seqi; mkdir -p gpurun_out/seqi
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seqi -- python3 tools/bench_infer.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
fn = glob.glob('gpurun_out/seqi/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
def name(r): return r['Kernel_Name'].replace('void hno::', '').replace('hno::', '').replace('void at::native::', 'at::')[:100]
idx = [i for i, r in enumerate(rows) if 'conv_k2s2_fwd' in r['Kernel_Name']]
print('markers', len(idx), 'kernels', len(rows))
a, b = idx[-2], idx[-1]
out, tot = [], 0.0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    out.append('%7.1f  %s' % (d, name(r)))
out.append('kernels %d  sum of durations %.1f us  span %.1f us' % (b - a, tot, (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3))
open('gpurun_out/seqi/sequence_infer.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
find gpurun_out/seqi -name "*kernel_trace.csv" -delete; find gpurun_out/seqi -name "*agent_info.csv" -delete
