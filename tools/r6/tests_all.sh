# full GPU suite + a short headline run: bash tools/r6/tests_all.sh [tag]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r06_t}
mkdir -p gpurun_out/$T
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "tests rc $?"; tail -6 gpurun_out/$T/tests.log
python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2> gpurun_out/$T/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['schedule'], d['config']['schedule_measured_ms'], d['config']['ms_per_step_bursts'])"
grep hno gpurun_out/$T/bench.err
