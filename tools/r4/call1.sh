# round 4, GPU call 1: full GPU suite on the fixed build, smoke, baseline bench, V-Net-DS bf16 kernel stats, attention baseline
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vnet -- python3 tools/dbg/vnet_ab.py 0 > $O/vnet.log 2>&1
find $O/vnet -name "*agent_info.csv" -delete; find $O/vnet -name "*kernel_trace.csv" -delete
python3 tools/dbg/hmha_one.py > $O/hmha.log 2>&1
python3 tools/bench_models.py hartleymha > $O/mha_model.log 2>&1
tail -3 $O/tests.log; tail -2 $O/smoke.log; tail -c 600 $O/bench.log; cat $O/vnet.log $O/hmha.log | tail -5
