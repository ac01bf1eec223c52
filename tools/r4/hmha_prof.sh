cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4h -- python3 tools/dbg/hmha_one.py > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/r4h/*/*kernel_stats.csv 1 12 | cut -c1-170
rm -rf gpurun_out/r4h
