# SQ counters of the kernels of one bench_models case whose name contains <pattern>: bash tools/r6/pmc_model.sh <case> <pattern>
CASE=${1:-hartleymha}; PAT=${2:-hmha3}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_model; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p1 -- python3 tools/bench_models.py $CASE > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 tools/bench_models.py $CASE > /dev/null 2>&1
python3 - "$O" "$PAT" <<'PY'
import csv, glob, collections, sys, json
o, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ('p1', 'p2'):
    for fn in glob.glob(f'{o}/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            n = r['Kernel_Name'].replace('void hno::', '').split('(')[0][:50]
            if pat in n: acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for fn in glob.glob(f'{o}/{d}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            n = r['Kernel_Name'].replace('void hno::', '').split('(')[0][:50]
            if pat in n: dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, cs in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    m['duration_us_under_pmc'] = sum(dur[n]) / max(len(dur[n]), 1)
    print(n, json.dumps({k: round(v, 1) for k, v in m.items()}))
PY
find $O -name "*.csv" -delete
