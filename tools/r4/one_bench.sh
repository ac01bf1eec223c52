cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile > gpurun_out/ob.log 2>&1; echo "rc $?"
grep -v Warning gpurun_out/ob.log | tail -15 | cut -c1-400
