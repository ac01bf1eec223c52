# Calibration of FETCH_SIZE / WRITE_SIZE on kernels with exactly known bytes and the pointwise kernels' access shapes
# (tools/dbg/planar_copy.bin: every kernel reads 2 x 48 x V floats and writes 2 x 24 x V floats): bash tools/dbg/pmc_planar.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_planar_f gpurun_out/pmc_planar_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_planar_f -- ./tools/dbg/planar_copy.bin > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_planar_w -- ./tools/dbg/planar_copy.bin > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, json
def read(d, counter):
    acc = collections.defaultdict(list)
    for fn in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if r['Counter_Name'] == counter:
                # grid size tells N = 65 from N = 64 apart only through the launch order: keep order per kernel name
                acc[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return acc
f, w = read('pmc_planar_f', 'FETCH_SIZE'), read('pmc_planar_w', 'WRITE_SIZE')
out = {}
for k in sorted(f):
    vals_f, vals_w = f[k], w.get(k, [])
    # the binary runs N = 65 first, then N = 64: first half / second half of each kernel's launches
    h = len(vals_f) // 2
    for tag, N, sl in (('N65', 65, slice(0, h)), ('N64', 64, slice(h, None))):
        V = N ** 3
        rb, wb = 2 * 48 * V * 4, 2 * 24 * V * 4
        ff = vals_f[sl]; ww = vals_w[sl] if vals_w else []
        if not ff: continue
        mf = sorted(ff)[len(ff) // 2] * 1024; mw = (sorted(ww)[len(ww) // 2] * 1024) if ww else float('nan')
        out[f'{k} {tag}'] = {'launches': len(ff), 'FETCH_SIZE_bytes': round(mf), 'read_bytes': rb, 'fetch_over_read': round(mf / rb, 3),
                             'WRITE_SIZE_bytes': round(mw) if mw == mw else None, 'written_bytes': wb, 'write_over_written': round(mw / wb, 3) if mw == mw else None}
json.dump(out, open('gpurun_out/pmc_planar.json', 'w'), indent=1)
for k, v in out.items(): print(k[:90], v['launches'], 'fetch/read', v['fetch_over_read'], 'write/written', v['write_over_written'])
PY
