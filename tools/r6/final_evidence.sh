# round-6 evidence from the current tree: bash tools/r6/final_evidence.sh <tag>     (everything lands under gpurun_out/<tag>)
TAG=${1:-r06_f}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err
grep '^{"metric"' $O/bench.log > $O/bench.json
STEPS=30; WARM=5
# per-kernel times: one pass over the batch on ONE stream (kernels do not overlap), then the default (measured) schedule
HNO_SPLIT_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- python3 bench.py --steps $STEPS --warmup $WARM --bursts 1 --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph.log 2>&1
HNO_SPLIT_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph_split -- python3 bench.py --steps $STEPS --warmup $WARM --bursts 1 --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph_split.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -- python3 bench.py --steps 3 --warmup 1 --bursts 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > $O/bench_eager.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 1 --bursts 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 1 --bursts 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
bash tools/dbg/pmc_step.sh > $O/pmc_step.log 2>&1; cp gpurun_out/pmc_step.json $O/pmc_sq_per_kernel.json
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
for sub in ('graph', 'graph_split'):
    fn = glob.glob(o + f'/{sub}/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(fn)))
    n = [int(r['Calls']) for r in rows if 'labels4' in r['Name']][0]      # one label conversion per step
    at = sum(int(r['Calls']) for r in rows if 'at::native' in r['Name'] and int(r['Calls']) >= n)
    red = {r['Name'].split('(')[0]: int(r['Calls']) / n for r in rows if 'reduce_partials' in r['Name']}
    tot = sum(int(r['Calls']) * float(r['AverageNs']) for r in rows) / n / 1e3
    open(o + f'/{sub}_steps.txt', 'w').write(f'{n} steps in the file (python tools/kstats.py <file> {n}); sum of kernel time per step {tot:.1f} us; reduce launches per step: {red}; '
                                             f'at::native kernels launched at least once per step: {at}\n')
    print(sub, open(o + f'/{sub}_steps.txt').read().strip())
fe = glob.glob(o + '/fetch/**/*counter_collection.csv', recursive=True)[0]
wr = glob.glob(o + '/write/**/*counter_collection.csv', recursive=True)[0]
open(o + '/pmc_files.txt', 'w').write(fe + '\n' + wr + '\n')
PY
python3 tools/make_hbm_traffic.py $(sed -n 1p $O/pmc_files.txt) $(sed -n 2p $O/pmc_files.txt) $O/r > $O/hbm_traffic.log 2>&1; cp profiles/hbm_traffic.json $O/hbm_traffic.json
# cfg3 with bf16 activations in memory on / off (same box), per-kernel stats of cfg3 bf16, HartleyMHASeg and V-Net-DS bf16
bash tools/r6/cfg3_ab.sh > $O/ab_cfg3_io16.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3 -o run -- python3 tools/bench_models.py fnoseg_cfg3:bf16 > $O/cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mha -o run -- python3 tools/bench_models.py hartleymha > $O/mha.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vnet -- python3 tools/dbg/vnet_ab.py 0 > $O/vnet.log 2>&1
python3 tools/bench_models.py fnoseg_cfg3 fnoseg_cfg3:bf16 hnoseg hartleymha vnetds_cfg4:bf16 vnetds_cfg4 fno_individual > $O/models.jsonl 2> /dev/null
python3 tools/bench_infer.py > $O/inference.jsonl 2>/dev/null; python3 tools/bench_infer.py --size 155 240 240 >> $O/inference.jsonl 2>/dev/null
# the round's experiments: complementary kernels side by side, the fused inverse + pointwise probe, the block tail with bf16 tensors,
# 64-byte against 128-byte row segments, the stagger
if [ -z "$SKIP_EXPERIMENTS" ]; then
python3 tools/r6/corun.py 2>/dev/null | tail -1 > $O/corun.jsonl
python3 tools/r6/invpw_lab.py 2>/dev/null | tail -1 > $O/invpw_probe.json
python3 tools/r6/branch_lab.py 2>/dev/null | tail -1 > $O/branch_lab.json
SEG_B=8 ./tools/r6/seg_bench.bin > $O/seg_bench.txt 2>&1
SPLIT=1 bash tools/r6/ab_env.sh "HNO_PW_STAGGER=0" "HNO_PW_STAGGER=2" > $O/ab_stagger.txt 2>&1
fi
# launch-ordered trace of one V-Net-DS cfg4 bf16 step (tools/r6/vnet_trace.sh)
bash tools/r6/vnet_trace.sh $TAG/vnet_trace > /dev/null 2>&1; cp gpurun_out/$TAG/vnet_trace/step_trace.txt $O/vnet_step_trace.txt; rm -rf gpurun_out/$TAG/vnet_trace/prof
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*domain_stats.csv" -delete
ls -R $O | head -70; tail -c 900 $O/bench.json
