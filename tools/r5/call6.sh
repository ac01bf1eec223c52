cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "spectral_middle or fourier_middle or batch_4 or channel_padded or plane_kernels or hnosegxs or noseg or small_models or benchmark_shapes or benched or dht_crop or residual or roundtrip or idht" 2>&1 | tail -3
HNO_MID_ZLAYOUT=0 timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "spectral_middle or fourier_middle or hnosegxs_128" 2>&1 | tail -2
HNO_MID_ZLAYOUT=2 timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "spectral_middle_vs_three or hnosegxs_128" 2>&1 | tail -2
bash tools/r5/ab_trees.sh "plain run|dht_|spec_mid|sum of kernel"
