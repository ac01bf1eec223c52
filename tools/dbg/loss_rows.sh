# usage (GPU box): bash tools/dbg/loss_rows.sh -- loss statistics / finalize kernel time against the number of per-sample rows
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 128 256 512 1024; do
  rm -rf gpurun_out/lossrows
  HNO_LOSS_ROWS=$r rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lossrows -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary > /dev/null 2>&1
  f=$(ls gpurun_out/lossrows/*/*kernel_stats.csv | head -1)
  echo "rows $r: $(python3 -c "
import csv,sys
for row in csv.DictReader(open('$f')):
    if 'loss_stats_vec' in row['Name'] or 'loss_finalize' in row['Name']: print(row['Name'][11:34], round(float(row['AverageNs'])/1e3,1), 'us', end='   ')
")"
done
