# usage (GPU box): bash tools/dbg/prof_mid.sh [size] -- kernel stats of the HNOSeg-XS step at another input size (generic-kernel grids)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/midprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/midprof -- python3 tools/dbg/midplane_ab.py ${1:-96} > gpurun_out/mid.log 2>&1
find gpurun_out/midprof -name "*agent_info.csv" -delete; find gpurun_out/midprof -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/mid.log
