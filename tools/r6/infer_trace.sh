# launch-ordered trace of one HNOSeg-XS inference at 240 x 240 x 155 (tools/bench_infer.py): bash tools/r6/infer_trace.sh [tag]
TAG=${1:-r06_it}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/prof -o run -- python3 tools/bench_infer.py --samples 3 > gpurun_out/$TAG/log.txt 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob('gpurun_out/$TAG/prof/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'argmax' in n or 'up_argmax' in n or 'uphead' in n]
seq = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(seq[0]['Start_Timestamp'])
tot = 0; prev = t0
out = open('gpurun_out/$TAG/step_trace.txt', 'w')
for r in seq:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (en - st) / 1e3; tot += d
    out.write('%8.1f %7.1f gap %6.1f  %-70s grid %sx%sx%s/%s\n' % ((st - t0) / 1e3, d, (st - prev) / 1e3, r['Kernel_Name'][:70], r.get('Grid_Size_X'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'), r.get('Workgroup_Size_X')))
    prev = en
out.write('launches %d, sum of durations %.1f us, span %.1f us\n' % (len(seq), tot, (int(seq[-1]['End_Timestamp']) - t0) / 1e3))
out.close()
print(open('gpurun_out/$TAG/step_trace.txt').read())
PY
