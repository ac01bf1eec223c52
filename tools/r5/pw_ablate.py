"""Upper bound of any matrix-core speed-up in the pointwise forward: the 48 -> 24 kernel at 2 x 65^3 (channel-padded rows) with its MFMA chain
replaced by adds (debug flag 1, results wrong) and without its epilogue + stores (flag 2), from a graph replay."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
L = pkg._lib.lib(); P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev = 'cuda'; B, C, V = 2, 24, 274656
xa = torch.randn(B, C, V, device=dev); xb = torch.randn_like(xa); yy = torch.empty_like(xa)
W = torch.randn(C, 2 * C, device=dev) * 0.1; bias = torch.randn(C, device=dev) * 0.01
big = torch.empty(64 * 1024 * 1024, device=dev)          # 256 MB: flushes the caches between replays


def timeit(fn, n=10, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for flags, name in ((0, 'full'), (1, 'adds instead of the MFMA chain'), (2, 'no epilogue, no stores'), (3, 'loads + LDS reads only')):
    L.hno_set_debug(flags)
    t = timeit(lambda: L.hno_pwconv_fwd(P(xa), 24, P(xb), 24, P(W), P(bias), P(yy), B, 24, V, 1, S()))
    print(f'pwconv_fwd 48 -> 24 [{name}]: {t:.1f} us')
L.hno_set_debug(0)
