cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-40s %.3f ms  %.1f vol/s' % ('$*', d['ms_per_step'], d['value']))"; }
run A=0
for g in 98 130 160 256; do run HNO_FWD_GRID=$g; done
run HNO_PWCHAIN_WAVES=8
run HNO_PWF_WAVES=12
run HNO_UPR_WGS=256
run HNO_UPR_WGS=1024
run A=1
