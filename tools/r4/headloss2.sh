cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "head or upsoftmax or loss or small_models or expected" 2>&1 | tail -5
for wgs in 512 1024; do
  HNO_UPR_WGS=$wgs python3 bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/hl_$wgs.log 2>&1; echo "rc $?"
  grep -v Warning gpurun_out/hl_$wgs.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('WGS=$wgs', d['value'], d['ms_per_step'], {n:v['avg_us'] for n,v in d['kernels'].items() if any(s in n for s in ('up','loss','label'))})
"
done
