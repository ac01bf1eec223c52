cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_hip_ops.py tests/test_training_loop.py -m gpu -q -k "hnosegxs or xsblock or channel_padded or training or batch_4 or small" 2>&1 | grep -E "passed|failed|FAILED|Error" | head
python3 __graft_entry__.py smoke 2>&1 | tail -2 | cut -c1-250
