cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
HNO_INV_PREFETCH=1 timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dht_crop_pad or residual_at_benchmark or hnosegxs_128 or spectral_middle_vs_float64" 2>&1 | tail -2
bash tools/r5/ab_env.sh HNO_INV_PREFETCH=0 HNO_INV_PREFETCH=1
