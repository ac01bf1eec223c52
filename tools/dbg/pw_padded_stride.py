"""Does a 128-byte aligned channel stride pay for the pointwise kernels?  Times hno_pwconv_fwd / hno_pwconv_bwd (48 -> 24, B = 2) at
V = 65^3 = 274 625 (every channel row at another 4-byte phase) and at V = 274 656 = 32 x 8 583 (the same rows padded to a multiple of
32 floats): same tile count, 0.011 % more bytes.   python tools/dbg/pw_padded_stride.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib()
P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev = 'cuda'


def timeit(fn, n=10, reps=5):
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


B, C = 2, 24
W = torch.randn(C, 2 * C, device=dev) * 0.1
bias = torch.randn(C, device=dev) * 0.01
for V in (65 ** 3, 274656):
    xa, xb = torch.randn(B, C, V, device=dev), torch.randn(B, C, V, device=dev)
    y = torch.empty(B, C, V, device=dev)
    t = timeit(lambda: L.hno_pwconv_fwd(P(xa), C, P(xb), C, P(W), P(bias), P(y), B, C, V, 1, S()))
    print(f'V={V}: pwconv_fwd 48->24: {t:.1f} us')
    gy = torch.randn_like(y); gxa, gxb = torch.empty_like(xa), torch.empty_like(xb)
    dW, db = torch.empty_like(W), torch.empty_like(bias)
    ws = torch.empty(L.hno_pwconv_bwd_workspace_bytes(48, 24) // 4, device=dev)
    t = timeit(lambda: L.hno_pwconv_bwd(P(gy), P(y), P(xa), C, P(xb), C, P(W), P(gxa), P(gxb), P(dW), P(db), P(ws), B, C, V, 1, 0, 0, S()))
    print(f'V={V}: pwconv_bwd 48->24 (+ slab reduce): {t:.1f} us')
    t = timeit(lambda: L.hno_pwconv_bwd(P(gy), P(y), P(xa), C, P(xb), C, P(W), P(gxa), P(gxb), P(dW), P(db), P(ws), B, C, V, 1, 1, 0, S()))
    print(f'V={V}: pwconv_bwd 48->24 with xa_act (+ slab reduce): {t:.1f} us')
