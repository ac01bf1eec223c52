# kernel sequence of one graph replay of a tools/bench_models.py model (launch order, durations): bash tools/r4/trace_seq_model.sh fnoseg_cfg3
M=${1:-fnoseg_cfg3}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/seqx; mkdir -p gpurun_out/seqx
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seqx -- python3 tools/bench_models.py $M > /dev/null 2>&1
python3 - "$M" <<'PY'
import csv, glob, sys
fn = glob.glob('gpurun_out/seqx/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
def name(r): return r['Kernel_Name'].replace('void hno::', '').replace('hno::', '').replace('void at::native::', 'at::')[:90]
idx = [i for i, r in enumerate(rows) if 'loss_finalize' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
out = []
tot = 0.0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    out.append('%7.1f  grid %-9s %s' % (d, r.get('Grid_Size', '?') + '/' + r.get('Workgroup_Size', '?'), name(r)))
out.append('kernels %d  sum of durations %.1f us' % (b - a, tot))
open('gpurun_out/seqx/sequence_%s.txt' % sys.argv[1], 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
find gpurun_out/seqx -name "*kernel_trace.csv" -delete; find gpurun_out/seqx -name "*agent_info.csv" -delete
