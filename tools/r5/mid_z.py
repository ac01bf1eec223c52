"""Timing experiment: the fused middle with the addressing of a workgroup-contiguous intermediate layout (debug flag 2048, results wrong)
against the shipped addressing; alone (intermediate resident in the XCD's L2 from the previous call) and behind the plane kernel that
writes the intermediate from other XCDs, as in the training step."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib()
P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev, N, B, C, modes = 'cuda', 65, 2, 24, (10, 14, 14)


def timeit(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


torch.manual_seed(0)
Ws = [torch.randn(C, C, device=dev) * 0.2 for _ in range(3)]
wp = ops._layer_ptrs(Ws)
x = torch.randn(B, C, N, N, N, device=dev)
ws = torch.randn(L.hno_dht3_workspace_bytes(B * C, N, N, N, *modes) // 4, device=dev)
zall = torch.randn((4, B, C, 20, 28, 28), device=dev)
dW = torch.empty(3, C, C, device=dev)
slab = torch.empty(L.hno_spec_mid_bwd_workspace_bytes(B, C, modes[1], 3) // 4, device=dev)
planes = lambda: L.hno_dht3_planes(P(x), P(ws), B * C, N, N, N, *modes, 0, S())
fw = lambda: L.hno_spec_mid_fwd(P(ws), wp, P(zall), B, C, N, *modes, 3, 1, 1, 1.0 / N ** 3, S())
bw = lambda: L.hno_spec_mid_bwd(P(ws), wp, P(zall), P(dW), P(slab), 4 * slab.numel(), B, C, N, *modes, 3, 1, 1, 1.0, S())
t_pl = timeit(planes)
print(f'plane kernel alone {t_pl:.1f} us')
for dbg in (0, 2048):
    L.hno_set_debug(dbg)
    print(f'dbg {dbg}: mid fwd alone {timeit(fw):.1f}  bwd alone {timeit(bw):.1f}  | behind the plane kernel: fwd {timeit(lambda: (planes(), fw())) - t_pl:.1f}  bwd {timeit(lambda: (planes(), bw())) - t_pl:.1f} us')
L.hno_set_debug(0)
