# per-launch trace of one graph-replayed V-Net-DS cfg4 bf16 step, in launch order: bash tools/r6/vnet_trace.sh [tag]
TAG=${1:-r06_vt}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/prof -o run -- python3 tools/bench_models.py vnetds_cfg4:bf16 > gpurun_out/$TAG/log.txt 2>&1
grep "^{" gpurun_out/$TAG/log.txt | cut -c1-300
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/$TAG/prof/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last replayed step: from the last labels/pack kernel backwards -- simply the last N launches where N = period of the sequence
names = [r['Kernel_Name'] for r in rows]
# find the period: the distance between the last two occurrences of the rarest hno kernel
cnt = collections.Counter(names)
anchor = 'hno::loss_finalize_kernel'
key = [n for n in cnt if n.startswith(anchor)][0]
idx = [i for i, n in enumerate(names) if n == key]
per = idx[-1] - idx[-2]
seq = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(seq[0]['Start_Timestamp'])
out = open('gpurun_out/$TAG/step_trace.txt', 'w')
tot = 0
for r in seq:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    out.write('%9.1f %8.1f  %-70s grid %s/%s lds %s\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, d, r['Kernel_Name'][:70], str(r.get('Grid_Size_X')) + 'x' + str(r.get('Grid_Size_Y')) + 'x' + str(r.get('Grid_Size_Z')), r.get('Workgroup_Size_X', r.get('Workgroup_Size')), r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', ''))))
out.write('launches %d, sum of durations %.1f us, span %.1f us\n' % (len(seq), tot, (int(seq[-1]['End_Timestamp']) - t0) / 1e3))
out.close()
print(open('gpurun_out/$TAG/step_trace.txt').read()[-400:])
PY
