cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "stem_chain or k2s2 or small_models" 2>&1 | tail -12
for sc in 1 0; do
  HNO_STEM_CHAIN=$sc python3 bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/sc_$sc.log 2>&1; echo "rc $?"
  grep -v Warning gpurun_out/sc_$sc.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('STEM_CHAIN=$sc', d['value'], d['ms_per_step'], {n:v['avg_us'] for n,v in d['kernels'].items() if any(s in n for s in ('k2s2','reduce'))})
"
done
