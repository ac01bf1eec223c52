# usage (GPU box): bash tools/dbg/prof_infer.sh  -- kernel stats of single-image inference at 240 x 240 x 155 (tools/bench_infer.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/inferprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inferprof -- python3 tools/bench_infer.py > gpurun_out/infer.log 2>&1
find gpurun_out/inferprof -name "*agent_info.csv" -delete; find gpurun_out/inferprof -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/infer.log | cut -c1-300
