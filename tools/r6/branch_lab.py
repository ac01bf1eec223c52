"""The fused block tail of FNOSeg under autocast, kernel by kernel (2 x 24 x 65^3, caches flushed between calls by a 512 MB fill):
forward (hno_pwconv_fwd_branch) and backward (hno_pwconv_bwd_branch) with fp32 and bf16 tensors in memory.
HNO_ALLOW_DEBUG_FLAGS=1 HNO_DEBUG_FLAGS=2 skips the backward's weight-gradient products (timing only: wrong weight gradients)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd._lib import lib, ptr, check, stream_ptr
L = lib()
B, C, N = 2, 24, 65
ld = ops._pad_ld(N ** 3)
def act(seed, rnd=True):
    t = ops.act_empty(B, C, (N, N, N), 'cuda', ld)
    t.as_strided((B * C * ld,), (1,)).normal_(generator=torch.Generator(device='cuda').manual_seed(seed))
    if rnd:
        t.copy_(t.bfloat16().float())
    return t
s, x, g = act(1, False), act(2), act(3)
Wbr, bbr = torch.randn(24, 24, device='cuda') * 0.2, torch.randn(24, device='cuda') * 0.1
W, b = torch.randn(24, 48, device='cuda') * 0.15, torch.randn(24, device='cuda') * 0.1
A = ops.ACT_SELU
x16, g16 = ops.to_bf16_layout(x, ld), ops.to_bf16_layout(g, ld)
y, o32 = ops.act_like(x), ops.act_like(x)
o16 = ops.act_empty16(B, C, (N, N, N), 'cuda', ld)
flush = torch.empty(128 << 20, device='cuda')
def fwd32(): check(L.hno_pwconv_fwd_branch(ptr(s), ptr(x), ptr(Wbr), ptr(bbr), ptr(W), ptr(b), ptr(y), ptr(o32), B, 24, 24, 24, ld, A | ops.ACT_BF16, stream_ptr()), 'f')
def fwd16(): check(L.hno_pwconv_fwd_branch(ptr(s), ptr(x16), ptr(Wbr), ptr(bbr), ptr(W), ptr(b), ptr(y), ptr(o16), B, 24, 24, 24, ld, A | ops.ACT_BF16 | ops.ACT_IO16, stream_ptr()), 'f')
def bwd32(): ops.pwconv_bwd_branch_raw(g, o32, y, x, W, Wbr, A, A, bf16=True)
def bwd16(): ops.pwconv_bwd_branch_raw(g16, o16, y, x16, W, Wbr, A, A, bf16=True, io16=True)
fwd32(); fwd16()
res = {}
for name, f in (('fwd_fp32_tensors', fwd32), ('fwd_bf16_tensors', fwd16), ('bwd_fp32_tensors', bwd32), ('bwd_bf16_tensors', bwd16)):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    res[name] = round(ts[len(ts) // 2], 1)
print(json.dumps({'flags': os.environ.get('HNO_DEBUG_FLAGS', '0'), 'us': res}))
