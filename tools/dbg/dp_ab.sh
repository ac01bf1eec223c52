# usage (GPU box): bash tools/dbg/dp_ab.sh  -- headline step vs the data-parallel path on one rank (eager / captured all-reduce), same box
cd $GRAFT_REPO_ROOT
F="--steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile"
for i in 1 2; do
  echo "normal:   $(python bench.py $F 2>&1 | grep "^{\|failed" | tail -1 | grep -o "\"ms_per_step\": [0-9.]*\|failed.*\|\"launch\": \"[^\"]*\"" | tr "\n" " ")"
  echo "dp path:  $(python bench.py --dp-path $F 2>&1 | grep "^{\|failed" | tail -1 | grep -o "\"ms_per_step\": [0-9.]*\|failed.*\|\"launch\": \"[^\"]*\"" | tr "\n" " ")"
  echo "captured: $(HNO_DP_CAPTURE_ALLREDUCE=1 timeout 300 python bench.py --dp-path $F 2>&1 | grep "^{\|failed" | tail -2 | grep -o "\"ms_per_step\": [0-9.]*\|failed.*\|\"launch\": \"[^\"]*\"" | tr "\n" " ")"
done
