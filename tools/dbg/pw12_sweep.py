import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib(); P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev = 'cuda'; B, N = 1, 65
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (Ca, Cb, Co) in ((12, 12, 12), (12, 0, 12), (12, 0, 4), (24, 24, 24)):
    xa = torch.randn(B, Ca, N, N, N, device=dev); xb = torch.randn(B, Cb, N, N, N, device=dev) if Cb else None
    W = torch.randn(Co, Ca + Cb, device=dev) * 0.1; bias = torch.randn(Co, device=dev) * 0.01
    y = ops.PwConvFn.apply(xa, xb, W, bias, ops.ACT_SELU)
    gy = torch.randn_like(y); gxa = torch.empty_like(xa); gxb = torch.empty_like(xb) if Cb else None
    dW, db = torch.empty_like(W), torch.empty_like(bias)
    ws = torch.empty(L.hno_pwconv_bwd_workspace_bytes(Ca + Cb, Co) // 4, device=dev)
    yy = torch.empty_like(y)
    tf = timeit(lambda: L.hno_pwconv_fwd(P(xa), Ca, P(xb), Cb, P(W), P(bias), P(yy), B, Co, N ** 3, 1, S()))
    for grid in (0, 256, 512, 1024):
        L.hno_set_debug(grid << 8)
        tb = timeit(lambda: L.hno_pwconv_bwd(P(gy), P(y), P(xa), Ca, P(xb), Cb, P(W), P(gxa), P(gxb), P(dW), P(db), P(ws), B, Co, N ** 3, 1, 0, 0, S()))
        print(f'{Ca}+{Cb}->{Co} B={B}: fwd {tf:.1f} us; bwd grid {grid or "default"}: {tb:.1f} us (incl. slab reduce)')
    L.hno_set_debug(0)
