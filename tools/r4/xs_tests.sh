cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_hip_ops.py tests/test_training_loop.py -m gpu -q -x -k "hnosegxs or xsblock or channel_padded or training or batch_4 or small" 2>&1 | tail -5
python3 __graft_entry__.py smoke 2>&1 | tail -2
python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
HNO_PW_CHAIN=0 python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
