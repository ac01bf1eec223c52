# per-kernel SQ counters of one eager bench step: bash tools/dbg/pmc_step.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_step1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d gpurun_out/pmc_step2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('pmc_step1', 'pmc_step2'):
    for fn in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            n = r['Kernel_Name'].replace('void hno::', '').replace('hno::', '').split('(')[0][:60]
            if n.startswith(('at::', '__amd')): continue
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for n, cs in acc.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    w = m.get('SQ_WAVES', 0) or 1
    out[n] = {'waves': round(w), 'wave_kcycles': round(4 * m.get('SQ_WAVE_CYCLES', 0) / w / 1e3, 1), 'wait_frac': round(m.get('SQ_WAIT_INST_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1), 2),
              'valu_per_wave': round(m.get('SQ_INSTS_VALU', 0) / w), 'salu_per_wave': round(m.get('SQ_INSTS_SALU', 0) / w), 'lds_per_wave': round(m.get('SQ_INSTS_LDS', 0) / w),
              'vmem_rd_per_wave': round(m.get('SQ_INSTS_VMEM_RD', 0) / w), 'vmem_wr_per_wave': round(m.get('SQ_INSTS_VMEM_WR', 0) / w),
              'lds_conflict_per_lds_active': round(m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_ACTIVE_INST_LDS', 1), 1), 2),
              'mfma_busy_frac_of_simd': round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(4 * m.get('SQ_WAVE_CYCLES', 1) / w, 1) , 3)}
json.dump(out, open('gpurun_out/pmc_step.json', 'w'), indent=1)
for n, v in sorted(out.items(), key=lambda kv: -kv[1]['wave_kcycles'] * kv[1]['waves'])[:14]:
    print(n, v)
PY
