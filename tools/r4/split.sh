cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for e in 1 0 1 0; do HNO_SPLIT_STREAMS=$e python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2>gpurun_out/split_$e.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('SPLIT=$e', d['value'], d['ms_per_step'], d['config']['schedule'], d['config']['final_loss'])"; done
grep -v "Warn\|amdgpu" gpurun_out/split_1.err | tail -5
