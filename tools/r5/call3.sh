cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for z in 1 0; do echo "== HNO_MID_ZLAYOUT=$z"; HNO_MID_ZLAYOUT=$z python3 tools/dbg/mid_lab.py 65 2>&1 | grep -E "^N=65: (z0|zero|unfused)|alone \[full\]"; HNO_MID_ZLAYOUT=$z python3 tools/dbg/mid_bwd_lab.py 2>&1 | grep -E "alone \[full\]"; done
timeout 1200 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "spectral_middle or fourier_middle or batch_4 or channel_padded or plane_kernels or hnosegxs or noseg or small_models or benchmark_shapes or benched" 2>&1 | tail -4
for z in 1 0; do HNO_MID_ZLAYOUT=$z HNO_SPLIT_STREAMS=0 python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zlayout', $z, d['value'], d['ms_per_step'], d['config']['schedule'])"; done
