cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r05_c}; mkdir -p gpurun_out/$T
python3 bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench.log 2> gpurun_out/$T/bench.err; echo "rc $?"
grep '^{"metric"' gpurun_out/$T/bench.log > gpurun_out/$T/bench.json
python3 - $T <<'PY'
import json, sys
d = json.load(open(f'gpurun_out/{sys.argv[1]}/bench.json'))
print(d['value'], d['ms_per_step'], d['config']['schedule'], d['config']['schedule_measured_ms'])
print('roofline', d['roofline'])
print('whole', d['whole_step_roofline'])
for k, v in d['secondary'].items(): print(' ', k, v)
print('cpu', d.get('cpu_baseline'))
for k, v in list(d['kernels'].items())[:14]: print(' ', k, v)
PY
grep hno gpurun_out/$T/bench.err | head -3
