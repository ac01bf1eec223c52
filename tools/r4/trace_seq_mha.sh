# kernel sequence of one attention block (forward and backward) inside a graph replay of the HartleyMHASeg step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/seqm; mkdir -p gpurun_out/seqm
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seqm -- python3 tools/bench_models.py hartleymha > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
fn = glob.glob('gpurun_out/seqm/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
def name(r): return r['Kernel_Name'].replace('void hno::', '').replace('hno::', '').replace('void at::native::', 'at::')[:84]
f = [i for i, r in enumerate(rows) if 'hmha2_kernel<0' in r['Kernel_Name']]
b2 = [i for i, r in enumerate(rows) if 'hmha2_kernel<2' in r['Kernel_Name']]
def dump(a, b, tag):
    t0 = int(rows[a]['Start_Timestamp']); prev = t0; out = [tag]
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        out.append('%8.1f us  dur %6.1f  gap %5.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name(r)))
        prev = e
    out.append('span us %.1f  kernels %d' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a))
    return out
# the last step: forward block between the 2nd-last and last forward attention launches of the final 16; backward likewise
out = dump(f[-3], f[-2], 'FORWARD block (attention launch to attention launch)') + dump(b2[-3], b2[-2], 'BACKWARD block')
open('gpurun_out/seqm/sequence.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
