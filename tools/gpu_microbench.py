"""Ablation timings of single kernels on the GPU (tuning aid; results with debug flags are wrong)."""
import sys, os, time, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib()
dev = 'cuda'

def timeit(fn, n=20, warm=3, reps=3):
    """GPU time per call from replaying a HIP graph of n back-to-back calls (a Python launch loop is
    host-bound below ~50 us per call)."""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

B, C, N = 2, 24, int(os.environ.get('MB_N', '65'))
xa = torch.randn(B, C, N, N, N, device=dev); xb = torch.randn_like(xa)
W = torch.randn(C, 2 * C, device=dev) * 0.1; bias = torch.randn(C, device=dev) * 0.01
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
if which in ('all', 'pwfwd'):
    for dbg, name in ((0, 'full'), (1, 'no mfma'), (2, 'no stores'), (3, 'loads only')):
        for grid in (0, 128, 256, 512, 1024):
            L.hno_set_debug(dbg | (grid << 8))
            yy = torch.empty_like(xa)
            PP, SS = pkg._lib.ptr, pkg._lib.stream_ptr
            t = timeit(lambda: L.hno_pwconv_fwd(PP(xa), 24, PP(xb), 24, PP(W), PP(bias), PP(yy), B, 24, N ** 3, 1, SS()))
            print(f'pwconv_fwd 48->24 [{name}] grid {grid or "default"}: {t:.1f} us  ({72 * 2 * N ** 3 * 4 / t / 1e3:.0f} GB/s algorithmic)')
    L.hno_set_debug(0)
    # plain copy reference: torch copy of the same bytes
    src = torch.randn(3 * 2 * 24 * N ** 3 // 2, device=dev); dst = torch.empty_like(src)
    t = timeit(lambda: dst.copy_(src)); print(f'torch copy {src.numel() * 8 / 1e6:.0f} MB moved: {t:.1f} us ({src.numel() * 8 / t / 1e3:.0f} GB/s)')
if which in ('all', 'pwbwd'):
    y = ops.PwConvFn.apply(xa, xb, W, bias, ops.ACT_SELU)
    gy = torch.randn_like(y)
    gxa, gxb = torch.empty_like(xa), torch.empty_like(xb)
    dW, db = torch.empty_like(W), torch.empty_like(bias)
    ws = torch.empty(L.hno_pwconv_bwd_workspace_bytes(48, 24) // 4, device=dev)
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    call = lambda: L.hno_pwconv_bwd(P(gy), P(y), P(xa), 24, P(xb), 24, P(W), P(gxa), P(gxb), P(dW), P(db), P(ws), B, 24, N ** 3, 1, 0, 0, S())
    for dbg, name in ((0, 'full nw12'), (64, 'full nw8'), (128, 'full nw4'), (64 | (512 << 8), 'nw8 grid512'), (128 | (512 << 8), 'nw4 grid512'), (128 | (1024 << 8), 'nw4 grid1024'),
                      (1, 'no dgrad mfma'), (2, 'no wgrad mfma'), (3, 'no mfma'), (4, 'no gx stores'), (7, 'loads+lds only')):
        L.hno_set_debug(dbg)
        t = timeit(call)
        print(f'pwconv_bwd 48->24 [{name}] (incl. slab reduce): {t:.1f} us ({316.4e6 / t / 1e3:.0f} GB/s algorithmic)')
    L.hno_set_debug(0)
if which in ('all', 'dhtab'):
    x = torch.randn(B, C, N, N, N, device=dev); z = torch.randn(B, C, 20, 28, 28, device=dev)
    BC, NN, mm = 48, (65, 65, 65), (10, 14, 14)
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, *NN, *mm) // 4, device=dev)
    out = torch.empty(B, C, 20, 28, 28, device=dev); yy = torch.empty_like(x)
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    for dbg, name in ((0, 'full'), (128, 'no role rotation'), (256, 'MFMA remainder tile'), (384, 'old assignment'), (0, 'full'), (384, 'old assignment'), (1, 'no stage W'), (2, 'no stage H'), (3, 'no W,H'), (7, 'load+fold only')):
        L.hno_set_debug(dbg)
        with pkg._lib.KernelProfile() as kp:
            for _ in range(20): L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, *NN, *mm, 1.0, S())
        print(f'dht_fwd_plane [{name}]: {kp.summary()["dht_fwd_plane_kernel"][2] * 1e3:.1f} us')
    for dbg, name in ((0, 'full'), (8, 'no SELU'), (1, 'no stage H'), (2, 'no stage W mma'), (4, 'no stores'), (6, 'no W mma, no stores'), (7, 'load only')):
        L.hno_set_debug(dbg)
        with pkg._lib.KernelProfile() as kp:
            for _ in range(20): L.hno_pad_idht3(P(z), None, 1, P(yy), P(ws), BC, *NN, *mm, 1.0, S())
        print(f'dht_inv_plane [{name}]: {kp.summary()["dht_inv_plane_kernel"][2] * 1e3:.1f} us')
    L.hno_set_debug(0)
if which in ('all', 'dht'):
    x = torch.randn(B, C, N, N, N, device=dev)
    z = torch.randn(B, C, 20, 28, 28, device=dev)
    with pkg._lib.KernelProfile() as kp:
        for _ in range(20):
            ops.dht3_crop_raw(x, (10, 14, 14), 1.0); ops.dht3_crop_raw(x, (10, 14, 14), 1.0, x, 1)
            ops.pad_idht3_raw(z, (N, N, N), 1.0, None, 1); ops.pad_idht3_raw(z, (N, N, N), 1.0, x, 0)
    for k, (c, s, avg) in kp.summary().items(): print(f'{k}: {avg * 1e3:.1f} us avg over {c}')
if which in ('stamps',):
    import ctypes
    x = torch.randn(B, C, N, N, N, device=dev)
    BC, NN, mm = 48, (65, 65, 65), (10, 14, 14)
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, *NN, *mm) // 4, device=dev)
    out = torch.empty(B, C, 20, 28, 28, device=dev)
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, *NN, *mm, 1.0, S())
    L.hno_set_debug(64)
    L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, *NN, *mm, 1.0, S())
    L.hno_set_debug(0)
    buf = (ctypes.c_longlong * 64)()
    L.hno_debug_stamps(buf, 64)
    st = list(buf)
    print(f'wall_clock64 (100 MHz) delta {st[61] - st[60]} -> {(st[61] - st[60]) / 100:.2f} us; clock64 delta {st[63] - st[62]} -> {(st[63] - st[62]) / max(1, st[61] - st[60]) * 100:.0f} MHz')
    print(f'block 0: start 0, end {(st[61]-st[60])/100:.2f} us; middle block: start {(st[56]-st[60])/100:.2f}, end {(st[57]-st[60])/100:.2f} us; last block: start {(st[58]-st[60])/100:.2f}, end {(st[59]-st[60])/100:.2f} us')
    print('fwd plane stamps (cycles since kernel-loop start; clock64 ticks):')
    names = ['iter top', 'staged', 'fetch issued', 'after W tiles', 'after leftover row', 'after H']
    for it in range(5):
        row = st[1 + it * 6: 7 + it * 6]
        if row[0] == 0: break
        print(f'  iter {it}:', ', '.join(f'{n}: {v - st[0]}' for n, v in zip(names, row)))

if which in ('stamps_dma',):
    import ctypes
    x = torch.randn(B, C, N, N, N, device=dev)
    BC, NN, mm = 48, (65, 65, 65), (10, 14, 14)
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, *NN, *mm) // 4, device=dev)
    out = torch.empty(B, C, 20, 28, 28, device=dev)
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    for _ in range(3): L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, *NN, *mm, 1.0, S())
    buf = (ctypes.c_longlong * 64)()
    L.hno_debug_stamps(buf, 64)   # allocate + clear
    L.hno_set_debug(64)
    L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, *NN, *mm, 1.0, S())
    L.hno_set_debug(0)
    L.hno_debug_stamps(buf, 64)
    st = list(buf)
    t0 = st[0]
    base = st[20]
    print(f'cycles from kernel top: table loads issued {st[21] - base}, DMAs issued + tables landed {st[22] - base}, end {st[23] - base}; wall: block 0 wave 0 {(st[61] - st[60]) * 10} ns, last block start +{(st[58] - st[60]) * 10} ns end +{(st[59] - st[60]) * 10} ns, middle block wave 7 start +{(st[56] - st[60]) * 10} end +{(st[57] - st[60]) * 10} ns')
    names = ['top', 'reads back (4 cos steps)', 'refill issued', 'sin loop done, next landed, cos prefetch issued', 'item done']
    for it in range(5):
        row = st[24 + it * 6: 29 + it * 6]
        if row[0] == 0: break
        print(f'  item {it}:', ', '.join(f'{n}: {v - base}' for n, v in zip(names, row)))

if which in ('stamps_item',):
    import ctypes
    z = torch.randn(B, C, 20, 28, 28, device=dev); x = torch.randn(B, C, N, N, N, device=dev)
    BC, NN, mm = 48, (65, 65, 65), (10, 14, 14)
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, *NN, *mm) // 4, device=dev)
    yy = torch.empty_like(x)
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    for addend in (None, x):
        for _ in range(3): L.hno_pad_idht3(P(z), P(addend), 1, P(yy), P(ws), BC, *NN, *mm, 1.0, S())
        buf = (ctypes.c_longlong * 64)()
        L.hno_debug_stamps(buf, 64)
        L.hno_set_debug(64)
        L.hno_pad_idht3(P(z), P(addend), 1, P(yy), P(ws), BC, *NN, *mm, 1.0, S())
        L.hno_set_debug(0)
        L.hno_debug_stamps(buf, 64)
        st = list(buf)
        base = st[20]
        print(f'addend={addend is not None}: cycles from kernel top: tables {st[21] - base}, end {st[23] - base}; wall: block 0 wave 0 {(st[61] - st[60]) * 10} ns, last block start +{(st[58] - st[60]) * 10} end +{(st[59] - st[60]) * 10} ns, middle block wave 11 start +{(st[56] - st[60]) * 10} end +{(st[57] - st[60]) * 10} ns')
        names = ['top', 'E landed + folds', 'H done', 'W done + stored to LDS', 'col 0 / row 0 done', 'epilogue done']
        for it in range(5):
            row = st[24 + it * 6: 30 + it * 6]
            if row[0] == 0: break
            print(f'  item {it}:', ', '.join(f'{n}: {v - base}' for n, v in zip(names, row)))

if which in ('stamps_inv',):
    import ctypes
    z = torch.randn(B, C, 20, 28, 28, device=dev); x = torch.randn(B, C, N, N, N, device=dev)
    BC, NN, mm = 48, (65, 65, 65), (10, 14, 14)
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, *NN, *mm) // 4, device=dev)
    yy = torch.empty_like(x)
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    for addend in (None, x):
        L.hno_pad_idht3(P(z), P(addend), 1, P(yy), P(ws), BC, *NN, *mm, 1.0, S())
        L.hno_set_debug(64)
        L.hno_pad_idht3(P(z), P(addend), 1, P(yy), P(ws), BC, *NN, *mm, 1.0, S())
        L.hno_set_debug(0)
        buf = (ctypes.c_longlong * 64)()
        L.hno_debug_stamps(buf, 64)
        st = list(buf)
        print(f'addend={addend is not None}: block 0 loop {(st[61] - st[60]) / 100:.2f} us, clock {(st[63] - st[62]) / max(1, st[61] - st[60]) * 100:.0f} MHz')
        print('  W detail (iter 1): reads done', st[40] - st[0], 'mfma done', st[41] - st[0], 'writes done', st[42] - st[0])
        names = ['iter top', 'staged', 'after H', 'chunk0 GEMM done', 'chunk0 epilogue done', '-']
        for it in range(5):
            row = st[1 + it * 6: 7 + it * 6]
            if row[0] == 0: break
            print(f'  iter {it}:', ', '.join(f'{n}: {v - st[0]}' for n, v in zip(names, row)))

if which in ('stamps_d',):
    import ctypes
    x = torch.randn(B, C, N, N, N, device=dev)
    BC, NN, mm = 48, (65, 65, 65), (10, 14, 14)
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, *NN, *mm) // 4, device=dev)
    out = torch.empty(B, C, 20, 28, 28, device=dev)
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, *NN, *mm, 1.0, S())
    L.hno_set_debug(64)
    L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, *NN, *mm, 1.0, S())
    L.hno_set_debug(0)
    buf = (ctypes.c_longlong * 64)()
    L.hno_debug_stamps(buf, 64)
    st = list(buf)
    print(f'D fwd kernel, block (3,5): loads issued->arrived {st[1]-st[0]} cycles, mfma {st[2]-st[1]}, store {st[3]-st[2]}; wall: this block {(st[13]-st[10])*10} ns; first block start -> last block end {(st[14]-st[11])*10} ns; last block start {(st[12]-st[11])*10} ns after first')
