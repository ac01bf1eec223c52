# the default bench.py invocation (headline + roofline + cpu_baseline + secondary), timed: bash tools/r5/bench_full.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
python3 bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "bench.py wall seconds: $(( $(date +%s) - T0 ))"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_full.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("schedule"))
for k, v in d["secondary"].items():
    if not isinstance(v, dict): print(" ", k, v)
print(d["cpu_baseline"])
PY
