# round-4 evidence from the final HEAD: bash tools/r4/final_evidence.sh <tag>
TAG=${1:-r04_f}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
grep '^{"metric"' $O/bench.log > $O/bench.json
STEPS=30; WARM=5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-kernel-profile --no-secondary > $O/bench_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > $O/bench_eager.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-kernel-profile --no-secondary > /dev/null 2>&1
bash tools/dbg/pmc_step.sh > $O/pmc_step.log 2>&1; cp gpurun_out/pmc_step.json $O/pmc_sq_per_kernel.json
# steps in the graph-replay file: every per-step kernel's Calls divided by its launches per step (loss_finalize: 1 per step)
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
fn = glob.glob(o + '/graph/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
n = [int(r['Calls']) for r in rows if 'labels' in r['Name'] and 'kernel' in r['Name']][0]      # (one label conversion per step; loss_finalize runs once per half-batch)
open(o + '/graph_steps.txt', 'w').write(f'{n} steps in kernel_stats_bench_graph_replay.csv (python tools/kstats.py <file> {n})\n')
print('graph steps', n)
PY
# V-Net-DS cfg4 bf16, HartleyMHASeg, every model family
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vnet -- python3 tools/dbg/vnet_ab.py 0 > $O/vnet.log 2>&1
python3 tools/dbg/hmha_one.py > $O/hmha.log 2>&1
python3 tools/bench_models.py fnoseg_cfg3 fnoseg_cfg3:bf16 hnoseg hartleymha vnetds_cfg4:bf16 vnetds_cfg4 fno_individual hnosegxs_cfg2@96 hnosegxs_cfg2@112 hnosegxs_cfg2@80 > $O/models.jsonl 2> /dev/null
python3 tools/bench_infer.py > $O/inference.jsonl 2>/dev/null
python3 tools/bench_cb_conv.py 0 > $O/bf16_conv_layers.txt 2>/dev/null
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
ls -R $O | head -50; cat $O/graph_steps.txt; tail -c 400 $O/bench.json
