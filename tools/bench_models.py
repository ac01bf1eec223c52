"""fwd + loss + bwd time of every model family at the BASELINE.json config sizes (parity-test configs, not
the headline bench line)."""
import sys, os, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.nets import custom_losses
nets = pkg.nets
pkg.ops.set_defer_reduce(os.environ.get('HNO_DEFER', '1') == '1')      # batched end-of-backward slab reductions, as bench.py / training()'s captured steps run them
CASES = {
    'hnosegxs_cfg2': (lambda: nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14)), (2, 4, 128, 128, 128)),
    'fnoseg_cfg3': (lambda: nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier'), (2, 4, 128, 128, 128)),
    'hnoseg': (lambda: nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Hartley'), (2, 4, 128, 128, 128)),
    'fno_individual': (lambda: nets.NeuralOperatorSeg(4, 4, 12, 4, (10, 14, 14), 'Fourier', weights_type='individual',
                                                      use_bias_conv_branch=True, use_block_skip=False), (2, 4, 128, 128, 128)),
    'hartleymha': (lambda: nets.HartleyMHASeg(4, 4, 12, 16, 4, (10, 14, 14), (2, 2, 2)), (1, 4, 128, 128, 128)),
    'vnetds_cfg4': (lambda: nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4]), (1, 4, 160, 192, 128)),
}
# a case name suffixed with ':bf16' runs the step under torch.autocast('cuda', dtype=torch.bfloat16) (the bf16 matrix-core path)
which = sys.argv[1:] or list(CASES)
import contextlib
import os
for name in which:
    bf16 = name.endswith(':bf16')
    ctor, shape = CASES[name.split(':')[0].split('@')[0]]
    if '@' in name:      # name@N: cubic inputs of edge N instead of the configuration's size (e.g. hnosegxs_cfg2@96)
        shape = tuple(shape[:2]) + (int(name.split(':')[0].split('@')[1]),) * 3
    ac = (lambda: torch.autocast('cuda', dtype=torch.bfloat16)) if bf16 else contextlib.nullcontext
    torch.manual_seed(0)
    model = ctor().cuda()
    x = torch.randn(shape, device='cuda')
    lab = pkg.ops.labels_prepare(torch.randint(0, 4, (shape[0], 1) + shape[2:], device='cuda').float(), 4)
    loss_fn = custom_losses.PCCLoss()
    def step():
        for p in model.parameters(): p.grad = None
        with ac():
            with pkg.ops.expected_loss(lab, loss_fn):
                y = model(x)
            loss = loss_fn(y, lab)
        loss.backward(); return loss
    try:
        step(); step()
        torch.cuda.synchronize(); t0 = time.time(); n = 5
        for _ in range(n): l = step()
        torch.cuda.synchronize(); dt = (time.time() - t0) / n
        lval = float(l.detach()); del l      # a live loss keeps last step's AccumulateGrad nodes (default stream) alive: that breaks the capture below
        # the same step replayed from a HIP graph (no host launch cost: what bench.py does for the headline config)
        dt_graph = None
        try:
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            gr.replay(); torch.cuda.synchronize(); t0 = time.time()
            for _ in range(10): gr.replay()
            torch.cuda.synchronize(); dt_graph = (time.time() - t0) / 10
        except Exception as e:
            print('graph capture failed:', repr(e)[:200], file=sys.stderr)
        # the schedule training() uses for the families that opted in (HNOSeg-XS): the two halves of the batch as two concurrent passes on
        # two streams of the graph (experiments.train_test.SampleSplit)
        dt_split = None
        try:
            from multimodal_3d_image_segmentation_amd.experiments.train_test import SampleSplit
            # measured for every shape of a family that opted in at some shape (the opt-in itself is per shape: this is how it was decided)
            forced = dict(os.environ, HNO_SPLIT_STREAMS='1') if getattr(model, 'hno_sample_split', False) else None
            prev_env = os.environ.get('HNO_SPLIT_STREAMS')
            if forced: os.environ['HNO_SPLIT_STREAMS'] = '1'
            try: can = (not bf16 or os.environ.get('HNO_SPLIT_STREAMS') == '1') and SampleSplit.usable(model, loss_fn, x)
            finally:
                if forced:
                    if prev_env is None: os.environ.pop('HNO_SPLIT_STREAMS', None)
                    else: os.environ['HNO_SPLIT_STREAMS'] = prev_env
            if can:
                sp = SampleSplit(model)
                def zero():
                    for p in model.parameters(): p.grad = None
                with torch.no_grad():
                    model(x[:shape[0] // 2])      # tables / kernel attributes of the half-batch shapes (not capturable)
                torch.cuda.synchronize()
                side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
                prev = pkg.ops.set_defer_reduce(True)
                try:
                    with torch.cuda.stream(side):
                        gs = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gs, stream=side, capture_error_mode='thread_local'):
                            sp.fwd_bwd(x, lab, loss_fn, zero_grad=zero, autocast=ac if bf16 else None)
                finally:
                    pkg.ops.set_defer_reduce(prev)
                torch.cuda.current_stream().wait_stream(side)
                gs.replay(); torch.cuda.synchronize(); t0 = time.time()
                for _ in range(10): gs.replay()
                torch.cuda.synchronize(); dt_split = (time.time() - t0) / 10
        except Exception as e:
            print('split capture failed:', repr(e)[:200], file=sys.stderr)
        with pkg._lib.KernelProfile() as kp:
            step()
        torch.cuda.synchronize()
        top = sorted(kp.summary().items(), key=lambda kv: -kv[1][1])[:8]
        print(json.dumps({'model': name, 'shape': shape, 'params': sum(p.numel() for p in model.parameters()),
                          'ms_per_step': round(dt * 1e3, 2), 'ms_per_step_graph': None if dt_graph is None else round(dt_graph * 1e3, 2),
                          'ms_per_step_graph_two_streams': None if dt_split is None else round(dt_split * 1e3, 2),
                          'volumes_per_s': round(shape[0] / (dt_graph or dt), 2), 'loss': round(lval, 5),
                          'max_mem_GB': round(torch.cuda.max_memory_allocated() / 1e9, 2),
                          'top_kernels_ms': {k: round(v[1], 2) for k, v in top},
                          'algorithmic_TFLOPs_or_TBps': {k: round(v[3] / (v[1] * 1e-3) / 1e12, 1) for k, v in top if v[3]}}))
    except Exception as e:
        print(json.dumps({'model': name, 'error': repr(e)[:300]}))
    del model, x
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
