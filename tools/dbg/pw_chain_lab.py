"""Chained pointwise layers (hno_pwconv_fwd_chain) against two hno_pwconv_fwd calls: values and time per call (graph replay)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib(); P, S = pkg._lib.ptr, pkg._lib.stream_ptr
torch.manual_seed(0)
B, C, n = 2, 24, 65
dev = 'cuda'
ld = ops._pad_ld(n ** 3)
mk = lambda: ops.to_layout(torch.randn(B, C, n, n, n, device=dev), ld)
u, t, k = mk(), mk(), mk()
Wc, Wm = torch.randn(C, 2 * C, device=dev) * 0.2, torch.randn(C, 2 * C, device=dev) * 0.2
bc, bm = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
xi_ref = ops.pwconv_fwd_raw(u, t, Wc, bc, ops.ACT_SELU)
xn_ref = ops.pwconv_fwd_raw(xi_ref, k, Wm, bm, ops.ACT_SELU)
xi, xn = ops.act_like(u), ops.act_like(u)
def chain():
    pkg._lib.check(L.hno_pwconv_fwd_chain(P(u), P(t), P(k), P(Wc), P(bc), P(Wm), P(bm), P(xi), P(xn), B, C, C, ld, ops.ACT_SELU, ops.ACT_SELU, S()), 'chain')
chain(); torch.cuda.synchronize()
print('xi equal', bool((xi == xi_ref).all()), 'xn max rel', float((xn - xn_ref).abs().max() / xn_ref.abs().max()))
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
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
def two():
    a = ops.pwconv_fwd_raw(u, t, Wc, bc, ops.ACT_SELU)
    return ops.pwconv_fwd_raw(a, k, Wm, bm, ops.ACT_SELU)
print(f'two layers {timeit(two):.1f} us, chained {timeit(chain):.1f} us')

# ---- backward
gn = mk()
def two_bwd():
    g_xi, g_k, dwm, dbm = ops.pwconv_bwd_raw(gn, xn_ref, xi_ref, k, Wm, ops.ACT_SELU, True)
    g_u, g_t, dwc, dbc = ops.pwconv_bwd_raw(g_xi, xi_ref, u, t, Wc, ops.ACT_SELU, True, xa_act=ops.ACT_SELU)
    return g_u, g_t, g_k, dwm, dbm, dwc, dbc
ref = two_bwd()
g_u, g_t, g_k = ops.act_like(u), ops.act_like(u), ops.act_like(u)
n1 = C * 2 * C + C
flat = torch.empty(2 * n1, device=dev)
ws = torch.empty(L.hno_pwconv_bwd_chain_workspace_bytes(C) // 4, device=dev)
def chain_bwd():
    pkg._lib.check(L.hno_pwconv_bwd_chain(P(gn), P(xn_ref), P(xi_ref), P(k), P(u), P(t), P(Wm), P(Wc), P(g_u), P(g_t), P(g_k), P(flat), P(ws),
                                          B, C, C, ld, ops.ACT_SELU, ops.ACT_SELU, ops.ACT_SELU, S()), 'chain bwd')
chain_bwd(); torch.cuda.synchronize()
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
# grads = [dWc | dbc | dWm | dbm]
print('gu', rel(g_u, ref[0]), 'gt', rel(g_t, ref[1]), 'gk', rel(g_k, ref[2]), 'dWm', rel(flat[n1:n1 + C * 2 * C].view(C, 2 * C), ref[3]), 'dbm', rel(flat[n1 + C * 2 * C:], ref[4]),
      'dWc', rel(flat[:C * 2 * C].view(C, 2 * C), ref[5]), 'dbc', rel(flat[C * 2 * C:n1], ref[6]))
print(f'backward: two layers {timeit(two_bwd):.1f} us, chained {timeit(chain_bwd):.1f} us')
