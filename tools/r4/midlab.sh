cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_hip_ops.py -m gpu -q -x -k "spectral_middle or batch_4 or fourier_middle" 2>&1 | tail -3
python3 tools/dbg/mid_lab.py 65 2>&1 | grep -E "alone \[full\]|3 layer|stamps|layer [0-9]:|ZL written|fused chain \[full\]"
python3 tools/dbg/mid_bwd_lab.py 2>&1 | grep -E "\[full\]" 
