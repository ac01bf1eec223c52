# usage (on the GPU box): bash tools/dbg/profile_round.sh <tag>   -- kernel stats (graph replay + eager) and the two PMC passes of bench.py
TAG=${1:-r02_c}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/graph -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-secondary > gpurun_out/$TAG/bench_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/eager -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > gpurun_out/$TAG/bench_eager.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
find gpurun_out/$TAG -name "*agent_info.csv" -delete
ls -R gpurun_out/$TAG | head -40
