cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_pwb1 -- python3 tools/dbg/pw_bwd_one.py 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_pwb2 -- python3 tools/dbg/pw_bwd_one.py 0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc_pwb1', 'pmc_pwb2'):
    acc = collections.defaultdict(list)
    for fn in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'pwconv_bwd_fast' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
