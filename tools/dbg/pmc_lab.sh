# per-kernel SQ counters of a lab script: bash tools/dbg/pmc_lab.sh <tag> <script> [args...]   (two PMC passes, kernel trace only)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_${TAG}_1 -- python3 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_${TAG}_2 -- python3 "$@" > /dev/null 2>&1
python3 - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in (f'pmc_{tag}_1', f'pmc_{tag}_2'):
    for fn in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            n = r['Kernel_Name'].replace('void hno::', '').replace('hno::', '').split('(')[0][:70]
            if n.startswith(('at::', '__amd', 'void at')): continue
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for n, cs in acc.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    w = m.get('SQ_WAVES', 0) or 1
    wc = max(m.get('SQ_WAVE_CYCLES', 1), 1)
    out[n] = {'waves': round(w), 'wave_kcycles': round(4 * wc / w / 1e3, 1), 'wait_any': round(m.get('SQ_WAIT_ANY', 0) / wc, 2),
              'wait_inst_any': round(m.get('SQ_WAIT_INST_ANY', 0) / wc, 2), 'active_inst_any': round(m.get('SQ_ACTIVE_INST_ANY', 0) / wc, 2),
              'valu/wave': round(m.get('SQ_INSTS_VALU', 0) / w), 'salu/wave': round(m.get('SQ_INSTS_SALU', 0) / w), 'smem/wave': round(m.get('SQ_INSTS_SMEM', 0) / w),
              'lds/wave': round(m.get('SQ_INSTS_LDS', 0) / w), 'vmem_rd/wave': round(m.get('SQ_INSTS_VMEM_RD', 0) / w), 'vmem_wr/wave': round(m.get('SQ_INSTS_VMEM_WR', 0) / w),
              'lds_conflict/lds_active': round(m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_ACTIVE_INST_LDS', 1), 1), 2),
              'mfma_busy_cycles/simd': round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024), 'sq_busy_cycles': round(m.get('SQ_BUSY_CYCLES', 0))}
json.dump(out, open(f'gpurun_out/pmc_{tag}.json', 'w'), indent=1)
for n, v in out.items():
    if 'dht' in n or 'spec' in n: print(n, v)
PY
