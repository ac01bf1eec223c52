cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o lab -- python3 /root/repo/tools/dbg/mid_lab.py 65 > /tmp/lab.log 2>&1
python3 - <<'PY'
import csv, glob
for fn in glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'dht' in r['Name'] or 'spec' in r['Name']:
            print(r['Name'][:80], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
