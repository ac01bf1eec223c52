cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/dbg/mid_lab.py 65 2>&1 | grep -E "^N=65: (z0|zero|unfused)|alone \[full\]|stamps|layer [0-9]|ZL"; python3 tools/dbg/mid_bwd_lab.py 2>&1 | grep -E "alone \[full\]|stamps"
timeout 1200 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "spectral_middle or batch_4 or hnosegxs or small_models or benched" 2>&1 | tail -3
