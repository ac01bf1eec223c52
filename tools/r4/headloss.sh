# fused head + loss: parity tests, then the headline with and without it (kernel profile on)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "head or upsoftmax or loss or small_models" 2>&1 | tail -15
for hl in 1 0; do
  HNO_HEAD_LOSS=$hl python3 bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/hl_$hl.log 2>&1; echo "rc $?"
  grep -v Warning gpurun_out/hl_$hl.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('HEAD_LOSS=$hl', d['value'], d['ms_per_step'])
for k in d.get('kernels', [])[:40]:
    if any(s in k['name'] for s in ('up','loss','label','head')): print('   ', k)
"
done
