cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for z in 2 1; do
O=gpurun_out/r05_z$z; rm -rf $O; mkdir -p $O
HNO_MID_ZLAYOUT=$z HNO_SPLIT_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
echo "== zlayout $z"; grep '^{"metric"' $O/bench_graph.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python3 - $O <<'PY'
import csv, glob, sys
fn = glob.glob(sys.argv[1] + '/graph/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(fn)))[:12]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
done
