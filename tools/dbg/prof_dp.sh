# usage (GPU box): bash tools/dbg/prof_dp.sh   -- kernel stats of the data-parallel step on one rank (bench.py --dp-path)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dp/graph -- python3 bench.py --dp-path --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-secondary > gpurun_out/dp/bench.log 2>&1
find gpurun_out/dp -name "*agent_info.csv" -delete
tail -1 gpurun_out/dp/bench.log | cut -c1-300
