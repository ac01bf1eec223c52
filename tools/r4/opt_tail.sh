cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "adamax or Adamax or optim or training or rccl or captured or head or loss or stem" 2>&1 | tail -6
python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | cut -c1-300
