cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg3prof -- python3 tools/bench_models.py fnoseg_cfg3 > /dev/null 2>&1
find gpurun_out/cfg3prof -name "*agent_info.csv" -delete; find gpurun_out/cfg3prof -name "*kernel_trace.csv" -delete
