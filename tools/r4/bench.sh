cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4b/bench.log 2>&1
python3 - <<'PY'
import json
for ln in open('gpurun_out/r4b/bench.log'):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); print(d['value'], d['ms_per_step'], d['config']['launch'], d['config']['host_us_per_step']); print(d['secondary'])
PY
tail -3 gpurun_out/r4b/bench.log | cut -c1-300
