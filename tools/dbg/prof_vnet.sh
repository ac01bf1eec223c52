cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vnetprof -- python3 tools/dbg/vnet_ab.py 0 > /dev/null 2>&1
find gpurun_out/vnetprof -name "*agent_info.csv" -delete; find gpurun_out/vnetprof -name "*kernel_trace.csv" -delete
