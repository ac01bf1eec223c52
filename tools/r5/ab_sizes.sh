cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for z in 1 0 1 0; do echo "HNO_MID_ZLAYOUT=$z"; HNO_MID_ZLAYOUT=$z python3 tools/bench_models.py hnosegxs_cfg2@96 hnosegxs_cfg2@112 hnosegxs_cfg2@80 hnosegxs_cfg2@64 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['model'], d['ms_per_step_graph'], {k: v for k, v in list(d['top_kernels_ms'].items())[:5]})"; done
