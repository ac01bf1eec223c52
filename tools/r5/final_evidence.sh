# round-5 evidence from the current HEAD: bash tools/r5/final_evidence.sh <tag>     (everything lands under gpurun_out/<tag>)
TAG=${1:-r05_f}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err
grep '^{"metric"' $O/bench.log > $O/bench.json
STEPS=30; WARM=5
# per-kernel times: one pass over the batch on ONE stream (kernels do not overlap), then the default (measured) schedule
HNO_SPLIT_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph_default -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph_default.log 2>&1
HNO_SPLIT_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph_split -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph_split.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > $O/bench_eager.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
bash tools/dbg/pmc_step.sh > $O/pmc_step.log 2>&1; cp gpurun_out/pmc_step.json $O/pmc_sq_per_kernel.json
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
for sub in ('graph', 'graph_default', 'graph_split'):
    fn = glob.glob(o + f'/{sub}/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(fn)))
    n = [int(r['Calls']) for r in rows if 'labels4' in r['Name']][0]      # one label conversion per step
    at = sum(int(r['Calls']) for r in rows if 'at::native' in r['Name'] and int(r['Calls']) >= n)
    red = {r['Name'].split('(')[0]: int(r['Calls']) / n for r in rows if 'reduce_partials' in r['Name']}
    open(o + f'/{sub}_steps.txt', 'w').write(f'{n} steps in the file (python tools/kstats.py <file> {n}); reduce launches per step: {red}; '
                                             f'at::native kernels launched at least once per step: {at}\n')
    print(sub, open(o + f'/{sub}_steps.txt').read().strip())
fe = glob.glob(o + '/fetch/**/*counter_collection.csv', recursive=True)[0]
wr = glob.glob(o + '/write/**/*counter_collection.csv', recursive=True)[0]
open(o + '/pmc_files.txt', 'w').write(fe + '\n' + wr + '\n')
PY
python3 tools/make_hbm_traffic.py $(sed -n 1p $O/pmc_files.txt) $(sed -n 2p $O/pmc_files.txt) $O/r > $O/hbm_traffic.log 2>&1; cp profiles/hbm_traffic.json $O/hbm_traffic.json
# same-box A/B of the intermediate layout (HNO_MID_ZLAYOUT) on the headline and on FNOSeg / HNOSeg
bash tools/r5/ab_env.sh HNO_MID_ZLAYOUT=1 HNO_MID_ZLAYOUT=0 > $O/ab_zlayout_headline.txt 2>&1
bash tools/r5/cfg3_ab.sh > $O/ab_zlayout_cfg3.txt 2>&1
# the other families
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vnet -- python3 tools/dbg/vnet_ab.py 0 > $O/vnet.log 2>&1
python3 tools/bench_models.py fnoseg_cfg3 fnoseg_cfg3:bf16 hnoseg hartleymha vnetds_cfg4:bf16 vnetds_cfg4 fno_individual hnosegxs_cfg2@96 hnosegxs_cfg2@112 hnosegxs_cfg2@80 > $O/models.jsonl 2> /dev/null
python3 tools/bench_infer.py > $O/inference.jsonl 2>/dev/null; python3 tools/bench_infer.py --size 155 240 240 >> $O/inference.jsonl 2>/dev/null
for m in fnoseg hnoseg vnetds; do python3 tools/bench_infer.py --model $m >> $O/inference.jsonl 2>/dev/null; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer -o run -- python3 tools/bench_infer.py --samples 8 > /dev/null 2>&1
# same-box A/B of the item plane kernels for general sizes (and of 65 x 65 through them): headline, other image sizes, inference
bash tools/r5/ab_env.sh HNO_ITEMS=1 HNO_ITEMS=0 > $O/ab_items_headline.txt 2>&1
bash tools/r5/ab_items.sh > $O/ab_items_sizes.txt 2>&1
python3 tools/dbg/mid_lab.py 65 2>&1 | grep -E "alone|stamps|layer [0-9]|ZL|fused chain|unfused" > $O/mid_lab.txt
python3 tools/dbg/mid_bwd_lab.py 2>&1 | grep -E "alone|stamps" >> $O/mid_lab.txt
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls -R $O | head -60; tail -c 600 $O/bench.json
