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
xa = torch.randn(B, C, N, N, N, device=dev); xb = torch.randn_like(xa)
W = torch.randn(C, 2 * C, device=dev) * 0.1; bias = torch.randn(C, device=dev) * 0.01
y = ops.PwConvFn.apply(xa, xb, W, bias, ops.ACT_SELU)
gy = torch.randn_like(y); gxa, gxb = torch.empty_like(xa), torch.empty_like(xb)
dW, db = torch.empty_like(W), torch.empty_like(bias)
ws = torch.empty(L.hno_pwconv_bwd_workspace_bytes(48, 24) // 4, device=dev)
for flags in [int(a) for a in sys.argv[1:]] or [0]:
    for grid in (0, 256, 512, 768):
        L.hno_set_debug(flags | (grid << 8))
        r = []
        for act in (1, 0):
            r.append(timeit(lambda: L.hno_pwconv_bwd(P(gy), P(y), P(xa), 24, P(xb), 24, P(W), P(gxa), P(gxb), P(dW), P(db), P(ws), B, 24, N ** 3, act, 0, 0, S())))
        print(f'flags {flags} grid {grid or "default"}: bwd 48->24 selu {r[0]:.1f} us, linear {r[1]:.1f} us (incl. slab reduce)')
L.hno_set_debug(0)
