"""Timing of the probe kernel (csrc/hno_invpw.hip: inverse plane transform inside the pointwise kernel, outputs NOT valid) against the
two kernels it would replace: hno_idht3_planes (+ SELU) then hno_pwconv_fwd (48 -> 24).  2 x 24 x 65^3, caches flushed between calls."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd._lib import lib, ptr, check, stream_ptr
L = lib()
B, C, N, modes = int(os.environ.get('LAB_B', '2')), 24, 65, (10, 14, 14)
ld = ops._pad_ld(N ** 3)
def act(seed):
    t = ops.act_empty(B, C, (N, N, N), 'cuda', ld)
    t.as_strided((B * C * ld,), (1,)).normal_(generator=torch.Generator(device='cuda').manual_seed(seed))
    return t
x, t = act(1), act(2)
W, b = torch.randn(24, 48, device='cuda') * 0.15, torch.randn(24, device='cuda') * 0.1
ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N, N, N, *modes) // 4, device='cuda')
check(L.hno_dht3_planes(ptr(x), ptr(ws), B * C, N, N, N, *modes, ld, stream_ptr()), 'planes')      # finite operands in the workspace
ws.mul_(1e-3)
u, xi = ops.act_like(x), ops.act_like(x)
TH, TW = torch.randn(65 * 16 * 2, device='cuda') * 0.1, torch.randn(32 * 65, device='cuda') * 0.1
flush = torch.empty(128 << 20, device='cuda')
def two_kernels():
    check(L.hno_idht3_planes(ptr(ws), None, ops.ACT_SELU, ptr(u), B * C, N, N, N, *modes, 1.0, ld, stream_ptr()), 'iplanes')
    check(L.hno_pwconv_fwd(ptr(u), 24, ptr(t), 24, ptr(W), ptr(b), ptr(xi), B, 24, ld, ops.ACT_SELU, stream_ptr()), 'pw')
def inv_only():
    check(L.hno_idht3_planes(ptr(ws), None, ops.ACT_SELU, ptr(u), B * C, N, N, N, *modes, 1.0, ld, stream_ptr()), 'iplanes')
def pw_only():
    check(L.hno_pwconv_fwd(ptr(u), 24, ptr(t), 24, ptr(W), ptr(b), ptr(xi), B, 24, ld, ops.ACT_SELU, stream_ptr()), 'pw')
def probe(grid):
    def f():
        check(L.hno_debug_invpw_fwd_probe(ptr(ws), ptr(t), ptr(W), ptr(b), ptr(TH), ptr(TW), ptr(u), ptr(xi), B, ld, 1.0, grid, stream_ptr()), 'probe')
    return f
res = {}
for name, f in (('inverse + SELU', inv_only), ('pointwise 48 -> 24', pw_only), ('both', two_kernels), ('probe grid 256', probe(256)), ('probe, no phase 1', probe(256 | (1 << 16))), ('probe, no axis-W products', probe(256 | (2 << 16))), ('probe, neither', probe(256 | (3 << 16))), ('probe, neither, u not stored', probe(256 | (7 << 16)))):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    res[name] = round(ts[len(ts) // 2], 1)
assert torch.isfinite(xi).all() and torch.isfinite(u).all()
print(json.dumps({'B': B, 'us (events, cold caches)': res}))
