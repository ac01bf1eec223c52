cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4v; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/dbg/vnet_ab.py 0 > $O/vnet.log 2>&1
python3 - <<'PY'
import csv, glob
fn = glob.glob('gpurun_out/r4v/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 548-ish kernels = the last graph replay: find the period by the pack kernel
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'cb_pack_weights_tiled' in n]
lo, hi = idx[-2], idx[-1]
with open('gpurun_out/r4v/one_step.csv', 'w') as f:
    t0 = int(rows[lo]['Start_Timestamp'])
    for r in rows[lo:hi]:
        f.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:.1f},{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.1f},{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']},{r['Kernel_Name'][:90]}\n")
print(hi - lo, 'kernels in one step', (int(rows[hi]['Start_Timestamp']) - t0) / 1e3, 'us')
PY
rm -rf $O/trace
