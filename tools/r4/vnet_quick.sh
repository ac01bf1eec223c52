cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
python3 -m pytest tests/test_bf16_ops.py tests/test_bf16_models.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
python3 tools/dbg/vnet_ab.py 0 ${VNET_FLAGS} 2>&1 | grep flag
