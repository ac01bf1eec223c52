cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_corun
for kv in "X=0" "HNO_PWF_WAVES=4 HNO_PWF_GRID_CAP=256 HNO_PWB_GRID_CAP=256 HNO_ITEM_FWD_WAVES=4 HNO_ITEM_INV_WAVES=6 HNO_ITEM_SMALL=1000000" "HNO_PWF_WAVES=4 HNO_PWF_GRID_CAP=256 HNO_PWB_GRID_CAP=256" "HNO_ITEM_FWD_WAVES=4 HNO_ITEM_INV_WAVES=6 HNO_ITEM_SMALL=1000000"; do
  env $kv python3 tools/r6/corun.py 2>&1 | tail -1 | tee -a gpurun_out/r06_corun/corun.jsonl
done
