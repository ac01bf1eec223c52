import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib(); P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev = 'cuda'; B, C, N = 2, 24, 65
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
s_in = torch.randn(B, C, N, N, N, device=dev); x = torch.randn_like(s_in)
Wbr = torch.randn(C, C, device=dev) * 0.1; bbr = torch.randn(C, device=dev) * 0.01
W = torch.randn(C, 2 * C, device=dev) * 0.1; bias = torch.randn(C, device=dev) * 0.01
y = torch.empty_like(x); out = torch.empty_like(x)
for grid in (0, 256, 512, 768):
    L.hno_set_debug(grid << 8)
    t = timeit(lambda: L.hno_pwconv_fwd_branch(P(s_in), P(x), P(Wbr), P(bbr), P(W), P(bias), P(y), P(out), B, 24, 24, 24, N ** 3, 1, S()))
    print(f'fwd_branch grid {grid or "default"}: {t:.1f} us')
L.hno_set_debug(0)
