cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_b; mkdir -p $O
python3 tools/dbg/mid_lab.py 65 2>&1 | grep -v Warning | tail -30
for s in 0 1; do HNO_SPLIT_STREAMS=$s python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split', $s, d['value'], d['ms_per_step'], d['config']['schedule'])"; done
timeout 600 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dhtn" 2>&1 | tail -3
