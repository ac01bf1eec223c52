# headline step time under single tuning knobs (same box, same call): bash tools/r4/knob_sweep.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-40s %.3f ms  %.1f vol/s' % ('$*', d['ms_per_step'], d['value']))"; }
run A=0
run A=1
run HNO_UPR_WGS=768
run HNO_UPR_WGS=1024
run HNO_UPR_WGS=384
run HNO_PWCHAIN_WAVES=8
run HNO_PWCHAIN_SLOTS=2
run HNO_PWF_WAVES=12
run HNO_FWD_GRID=512
run HNO_FWD_GRID=384
run A=2
