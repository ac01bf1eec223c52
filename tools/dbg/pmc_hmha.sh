cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/dbg/hmha_one.py
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_hmha -- python3 tools/dbg/hmha_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_hmha2 -- python3 tools/dbg/hmha_one.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc_hmha', 'pmc_hmha2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'hmha' in r['Kernel_Name']:
                mode = r['Kernel_Name'].split('<')[1].split(',')[0]
                acc[mode][r['Counter_Name']].append(float(r['Counter_Value']))
    for mode, cs in sorted(acc.items()):
        print(d, 'MODE', mode, {k: round(sum(v) / len(v)) for k, v in cs.items()})
PY
