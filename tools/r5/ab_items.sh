# same-box A/B of the item plane kernels for general sizes (HNO_ITEMS=0: the specialised / generic kernels of rounds 1-3)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for z in 1 0 1 0; do echo "HNO_ITEMS=$z"; HNO_ITEMS=$z python3 tools/bench_models.py hnosegxs_cfg2@80 hnosegxs_cfg2@96 hnosegxs_cfg2@112 hnosegxs_cfg2@144 hnosegxs_cfg2@160 hnosegxs_cfg2@192 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['model'], d['ms_per_step_graph'], {k: v for k, v in list(d['top_kernels_ms'].items())[:6]})"; done
for z in 1 0 1 0; do echo "HNO_ITEMS=$z"; HNO_ITEMS=$z python3 tools/bench_infer.py --samples 8 | cut -c100-260; done
