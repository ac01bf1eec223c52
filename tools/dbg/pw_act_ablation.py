"""pointwise forward/backward: activation on / off (how much of the kernel is the SELU epilogue?)"""
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
yy = torch.empty_like(xa)
for act in (1, 0):
    t = timeit(lambda: L.hno_pwconv_fwd(P(xa), 24, P(xb), 24, P(W), P(bias), P(yy), B, 24, N ** 3, act, S()))
    print(f'pwconv_fwd 48->24 act={act}: {t:.1f} us')
    t = timeit(lambda: L.hno_pwconv_fwd(P(xa), 24, None, 0, P(W[:, :24].contiguous()), P(bias), P(yy), B, 24, N ** 3, act, S()))
    print(f'pwconv_fwd 24->24 act={act}: {t:.1f} us')
y = ops.PwConvFn.apply(xa, xb, W, bias, ops.ACT_SELU)
gy = torch.randn_like(y); gxa, gxb = torch.empty_like(xa), torch.empty_like(xb)
dW, db = torch.empty_like(W), torch.empty_like(bias)
ws = torch.empty(L.hno_pwconv_bwd_workspace_bytes(48, 24) // 4, device=dev)
for act in (1, 0):
    t = timeit(lambda: L.hno_pwconv_bwd(P(gy), P(y), P(xa), 24, P(xb), 24, P(W), P(gxa), P(gxb), P(dW), P(db), P(ws), B, 24, N ** 3, act, 0, 0, S()))
    print(f'pwconv_bwd 48->24 act={act}: {t:.1f} us')
