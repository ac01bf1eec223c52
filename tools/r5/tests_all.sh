cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_t
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > gpurun_out/r05_t/tests.log 2>&1; echo "tests rc $?"; tail -6 gpurun_out/r05_t/tests.log
python3 bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2> gpurun_out/r05_t/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['schedule'], d['config']['schedule_measured_ms'])"
grep hno gpurun_out/r05_t/bench.err
