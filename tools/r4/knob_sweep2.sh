cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-40s %.3f ms  %.1f vol/s' % ('$*', d['ms_per_step'], d['value']))"; }
run A=0
for g in 256 240 224 208 195 130; do run HNO_FWD_GRID=$g; done
run A=1
