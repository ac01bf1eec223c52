cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for f in 0 16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_bwd_${f}_$c -- python3 tools/dbg/pw_bwd_one.py $f > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, glob
for f in (0, 16):
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = glob.glob(f'gpurun_out/pmc_bwd_{f}_{c}/**/*counter_collection.csv', recursive=True)
        vals = []
        for fn in files:
            for r in csv.DictReader(open(fn)):
                if 'pwconv_bwd_fast' in r['Kernel_Name'] and r['Counter_Name'] == c:
                    vals.append(float(r['Counter_Value']))
        print(f, c, len(vals), [round(v / 1024, 1) for v in vals[-5:]], 'MiB (raw counter is KB)')
PY
