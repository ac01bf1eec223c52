cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
python3 -m pytest tests/test_bf16_ops.py tests/test_bf16_models.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
env $A python3 tools/dbg/vnet_ab.py 0 2>&1 | grep flag | sed "s/^/$A /"
env $B python3 tools/dbg/vnet_ab.py 0 2>&1 | grep flag | sed "s/^/$B /"
done
