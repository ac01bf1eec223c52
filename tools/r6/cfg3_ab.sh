# cfg3 (FNOSeg 2 x 4 x 128^3, autocast bf16): bf16 activations in memory on / off, same box; then the new tests
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_cfg3
timeout 900 python3 -m pytest tests/test_bf16_ops.py -x -q -m gpu -k "bf16_planes or bf16_tensors or bf16_activations" 2>&1 | tail -15
for rep in 1 2; do
for io in 1 0; do
  HNO_IO16=$io python3 tools/bench_models.py fnoseg_cfg3:bf16 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('IO16=$io', d.get('ms_per_step_graph'), d.get('ms_per_step'), d.get('top_kernels_ms'), d.get('error'))"
done; done
