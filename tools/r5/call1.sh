# round 5, first GPU call: the new tests (schedule measurement, benched step vs golden, training loop) + a bench line + kernel trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_a; mkdir -p $O
timeout 900 python3 -m pytest tests/test_training_loop.py -x -q -m gpu > $O/tests_train.log 2>&1; echo "train tests rc $?"; tail -5 $O/tests_train.log
timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "benched_step or hnosegxs_128 or deferred" > $O/tests_bench.log 2>&1; echo "benched tests rc $?"; tail -5 $O/tests_bench.log
python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc $?"
grep '^{"metric"' $O/bench.log > $O/bench.json; grep -v Warning $O/bench.log | grep -v '^{"metric"' | tail -5
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05_a/bench.json'))
print(d['value'], d['ms_per_step'], d['config']['schedule'], d['config'].get('schedule_measured_ms'), d['roofline'])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob
fn = glob.glob('gpurun_out/r05_a/graph/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(fn)))[:40]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
