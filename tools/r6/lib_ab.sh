# two builds of libhno.so on one box, interleaved: bash tools/r6/lib_ab.sh <bench_models case> [libA libB ...]
# (build the other one with `git stash; make -C .../csrc; cp .../libhno.so .../libhno_base.so; git stash pop; make -C .../csrc`)
CASE=${1:-vnetds_cfg4:bf16}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/multimodal-3d-image-segmentation_amd
LIBS=${@:-"$P/libhno_base.so $P/libhno.so"}
for rep in 1 2 3; do
for lib in $LIBS; do
  HNO_LIB=$lib python3 tools/bench_models.py $CASE 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib)', d.get('ms_per_step_graph'), d.get('ms_per_step'), d.get('loss'), d.get('error'))"
done; done
