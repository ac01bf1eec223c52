# launch-ordered trace of one replayed headline step (one-pass schedule), with grids: bash tools/r6/headline_trace.sh [tag]
TAG=${1:-r06_ht}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
HNO_SPLIT_STREAMS=${SPLIT:-0} rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/prof -o run -- python3 bench.py --steps 10 --warmup 3 --bursts 1 --no-cpu-baseline --no-kernel-profile --no-secondary > gpurun_out/$TAG/log.txt 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob('gpurun_out/$TAG/prof/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'adamax_multi_dev' in n]
seq = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(seq[0]['Start_Timestamp'])
out = open('gpurun_out/$TAG/step_trace.txt', 'w')
tot = 0; prev_end = t0
for r in seq:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (en - st) / 1e3; tot += d
    out.write('%8.1f %7.1f gap %5.1f  %-64s grid %s x %s x %s / %s lds %s vgpr %s\n' % ((st - t0) / 1e3, d, (st - prev_end) / 1e3, r['Kernel_Name'][:64], r.get('Grid_Size_X'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'), r.get('Workgroup_Size_X'), r.get('LDS_Block_Size'), r.get('VGPR_Count')))
    prev_end = en
out.write('launches %d, sum of durations %.1f us, span %.1f us\n' % (len(seq), tot, (int(seq[-1]['End_Timestamp']) - t0) / 1e3))
out.close()
print(open('gpurun_out/$TAG/step_trace.txt').read())
PY
