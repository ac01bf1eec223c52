# HBM fetch of the chained stem backward with and without the XCD-aware tile order (PMC pass only: no trace domains): bash tools/r6/k2_xcd_pmc.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_k2xcd; mkdir -p $O
export HNO_K2_XCD=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/off -- python3 bench.py --steps 3 --warmup 1 --bursts 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
export HNO_K2_XCD=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/on -- python3 bench.py --steps 3 --warmup 1 --bursts 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
python3 - <<PY
import csv, glob
for tag in ('off', 'on'):
    f = sorted(glob.glob('$O/%s/**/*counter_collection.csv' % tag, recursive=True))[-1]
    n, v = 0, 0.0
    for r in csv.DictReader(open(f)):
        if 'conv_k2s2_chain_bwd' in r['Kernel_Name'] and r.get('Counter_Name') == 'FETCH_SIZE':
            n += 1; v += float(r['Counter_Value'])
    print('HNO_K2_XCD', tag, 'launches', n, 'FETCH_SIZE KB per launch', round(v / max(n, 1), 1), '-> bytes (x2 on gfx950, x1024)', round(2 * 1024 * v / max(n, 1) / 1e6, 1), 'MB')
PY
