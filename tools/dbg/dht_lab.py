"""Transform lab: correctness of hno_dht3_crop / hno_pad_idht3 against the float64 dense formulation (oracle/) on a few (b, c)
slabs, and GPU time per call from a HIP-graph replay (plane kernel + D kernel together; per-kernel split: rocprofv3 --kernel-trace).
Kernel variants are selected by environment variables read once per process (HNO_FWD_PLANE, HNO_INV_PLANE): run once per variant.

    python tools/dbg/dht_lab.py [N] [check|time|all]
"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from oracle import hno_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65
what = sys.argv[2] if len(sys.argv) > 2 else 'all'
B, C = 2, 24
modes = (10, 14, 14)
modes = tuple(min(m, N // 2) for m in modes)
dev = 'cuda'
tag = f"fwd={os.environ.get('HNO_FWD_PLANE', 'default')} inv={os.environ.get('HNO_INV_PLANE', 'default')}"


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def timeit(fn, n=20, warm=3, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


torch.manual_seed(0)
x = torch.randn(B, C, N, N, N, device=dev)
z = torch.randn(B, C, 2 * modes[0], 2 * modes[1], 2 * modes[2], device=dev)
add = torch.randn(B, C, N, N, N, device=dev)
if what in ('check', 'all'):
    # odd base offsets exercise the 16-byte alignment handling of the DMA path (views 1..3 floats into a larger buffer)
    for shift in (0, 1, 2, 3):
        buf = torch.randn(B * C * N ** 3 + 8, device=dev)
        xs = buf[shift:shift + B * C * N ** 3].view(B, C, N, N, N)
        y = ops.dht3_crop_raw(xs, modes, 1.0 / N ** 3)
        sel = [(0, 0), (1, C - 1), (0, 7)]
        err = max(rel(y[b, c].cpu(), O.dht_crop_dense(xs[b, c].cpu().double()[None, None], modes)[0, 0]) for b, c in sel)
        print(f'[{tag}] N={N} crop shift {shift}: rel err {err:.3e}  nan {int(torch.isnan(y).sum())}')
    out = ops.pad_idht3_raw(z, (N, N, N), 1.0)
    err = max(rel(out[b, c].cpu(), O.pad_idht_dense(z[b, c].cpu().double()[None, None], (N, N, N))[0, 0]) for b, c in [(0, 0), (1, C - 1)])
    print(f'[{tag}] N={N} pad: rel err {err:.3e}  nan {int(torch.isnan(out).sum())}')
    out = ops.pad_idht3_raw(z, (N, N, N), 0.5, add, ops.ACT_SELU)
    ref = torch.nn.functional.selu(0.5 * O.pad_idht_dense(z[1, 3].cpu().double()[None, None], (N, N, N))[0, 0] + add[1, 3].cpu().double())
    print(f'[{tag}] N={N} pad+add+selu: rel err {rel(out[1, 3].cpu(), ref):.3e}')
    # run-to-run determinism
    y1 = ops.dht3_crop_raw(x, modes, 1.0)
    bad = 0
    for _ in range(20):
        bad += int(not torch.equal(ops.dht3_crop_raw(x, modes, 1.0), y1))
    o1 = ops.pad_idht3_raw(z, (N, N, N), 1.0, add, ops.ACT_SELU)
    for _ in range(20):
        bad += int(not torch.equal(ops.pad_idht3_raw(z, (N, N, N), 1.0, add, ops.ACT_SELU), o1))
    print(f'[{tag}] repeat mismatches: {bad}')
if what in ('time', 'all'):
    L = pkg._lib.lib()
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    BC = B * C
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, N, N, N, *modes) // 4, device=dev)
    out = torch.empty_like(z)
    yy = torch.empty_like(x)
    t = timeit(lambda: L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, N, N, N, *modes, 1.0, S()))
    print(f'[{tag}] N={N} dht3_crop (plane + D): {t:.2f} us')
    t = timeit(lambda: L.hno_pad_idht3(P(z), None, 1, P(yy), P(ws), BC, N, N, N, *modes, 1.0, S()))
    print(f'[{tag}] N={N} pad_idht3 selu (D + plane): {t:.2f} us')
    t = timeit(lambda: L.hno_pad_idht3(P(z), P(add), 1, P(yy), P(ws), BC, N, N, N, *modes, 1.0, S()))
    print(f'[{tag}] N={N} pad_idht3 + addend + selu: {t:.2f} us')
if what == 'stream':
    # streaming floor of the DMA plane kernel: debug flag 1 = DMA + waits only (results are wrong)
    L = pkg._lib.lib()
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    BC = B * C
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, N, N, N, *modes) // 4, device=dev)
    out = torch.empty_like(z)
    for dbg in (0, 1, 2, 4, 0):
        L.hno_set_debug(dbg)
        t = timeit(lambda: L.hno_dht3_crop(P(x), None, 0, P(out), P(ws), BC, N, N, N, *modes, 1.0, S()))
        print(f'[{tag}] N={N} dht3_crop (plane + D), debug {dbg}: {t:.2f} us')
    L.hno_set_debug(0)
if what == 'invabl':
    L = pkg._lib.lib()
    P, S = pkg._lib.ptr, pkg._lib.stream_ptr
    BC = B * C
    ws = torch.empty(L.hno_dht3_workspace_bytes(BC, N, N, N, *modes) // 4, device=dev)
    yy = torch.empty_like(x)
    for dbg, act, ad, name in ((0, 1, None, 'selu'), (12, 1, None, 'no epilogue, no MFMA'), (8, 1, None, 'no MFMA'), (0, 0, None, 'no activation'), (2, 1, None, 'selu, no stores'), (4, 1, None, 'no epilogue'),
                               (0, 1, add, 'addend + selu'), (0, 0, add, 'addend, no activation'), (2, 1, add, 'addend + selu, no stores')):
        L.hno_set_debug(dbg)
        t = timeit(lambda: L.hno_pad_idht3(P(z), P(ad), act, P(yy), P(ws), BC, N, N, N, *modes, 1.0, S()))
        print(f'[{tag}] N={N} pad_idht3 (D + plane) [{name}]: {t:.2f} us')
    L.hno_set_debug(0)
