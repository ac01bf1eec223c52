import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
L = pkg._lib.lib(); P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev = 'cuda'; BZ, C, T = 4, 96, 1960
q = torch.randn(BZ, C, T, device=dev) * 0.1; k = torch.randn_like(q) * 0.1; v = torch.randn_like(q); do = torch.randn_like(q)
out = torch.empty_like(q); dq = torch.empty_like(q); dk = torch.empty_like(q); dv = torch.empty_like(q)
ws = torch.empty(L.hno_hmha_workspace_bytes(BZ, C, C, T) // 4, device=dev); WB = 4 * ws.numel()
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for flag in (1 << 24, 0):
    L.hno_set_debug(flag)
    tf = timeit(lambda: L.hno_hmha_fwd(P(q), P(k), P(v), P(out), P(ws), WB, BZ, C, C, T, 0.1, 1, S()))
    tb = timeit(lambda: L.hno_hmha_bwd(P(q), P(k), P(v), P(do), P(dq), P(dk), P(dv), P(ws), WB, BZ, C, C, T, 0.1, 1, S()))
    print('round-2 kernels' if flag else 'shared-tile kernels', end=': ')
fl = 2.0 * BZ * T * T * 2 * C
print(f'hmha fwd {tf:.1f} us ({fl / tf / 1e6:.1f} TFLOP/s), bwd {tb:.1f} us ({fl * 3.5 / tb / 1e6:.1f} TFLOP/s)')
# round 4b: the stream splits' partial results left to the consumer (ops.GroupedAttentionFn: the ungrouping permutation adds them)
L.hno_set_debug(0)
ns = L.hno_hmha_nsplit(BZ, T)
po = torch.empty(ns, BZ, C, T, device=dev); pq = torch.empty_like(po); pk = torch.empty_like(po); pv = torch.empty_like(po)
tf = timeit(lambda: L.hno_hmha_fwd_parts(P(q), P(k), P(v), P(po), BZ, C, C, T, 0.1, 1, S()))
tb = timeit(lambda: L.hno_hmha_bwd_parts(P(q), P(k), P(v), P(do), P(pq), P(pk), P(pv), BZ, C, C, T, 0.1, 1, S()))
print(f'partials form ({ns} splits, summed by the consumer): hmha fwd {tf:.1f} us ({fl / tf / 1e6:.1f} TFLOP/s), bwd {tb:.1f} us ({fl * 3.5 / tb / 1e6:.1f} TFLOP/s)')
