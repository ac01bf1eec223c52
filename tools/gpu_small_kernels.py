"""GPU time of the small latency-bound kernels (spectral mix, D-axis transforms) measured by replaying a
HIP graph of N back-to-back launches (no host launch overhead in the number)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib()
dev = 'cuda'

def graph_time(fn, n=50, reps=5):
    fn(); torch.cuda.synchronize()
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

B, C = 2, 24
z0 = torch.randn(B, C, 20, 28, 28, device=dev)
Ws = [torch.randn(C, C, device=dev) * 0.1 for _ in range(3)]
g = torch.randn_like(z0)
zs = ops.specmix_fwd_raw(z0, Ws, 1, ops.ACT_SELU)
for dbg in (0, 16):
    L.hno_set_debug(dbg)
    print(f'specmix fwd dbg={dbg}: {graph_time(lambda: ops.specmix_fwd_raw(z0, Ws, 1, ops.ACT_SELU)):.2f} us')
    print(f'specmix bwd (+reduce) dbg={dbg}: {graph_time(lambda: ops.specmix_bwd_raw(g, z0, zs, Ws, 1, ops.ACT_SELU)):.2f} us')
L.hno_set_debug(0)
x = torch.randn(B, C, 65, 65, 65, device=dev)
print(f'dht3_crop (plane + D): {graph_time(lambda: ops.dht3_crop_raw(x, (10, 14, 14), 1.0)):.2f} us')
print(f'pad_idht3 (D + plane): {graph_time(lambda: ops.pad_idht3_raw(z0, (65, 65, 65), 1.0, None, ops.ACT_SELU)):.2f} us')
y = torch.empty_like(z0)
print(f'torch add 3 MB: {graph_time(lambda: torch.add(z0, g, out=y)):.2f} us')
small = torch.randn(1024, device=dev); o = torch.empty_like(small)
print(f'torch add 4 KB: {graph_time(lambda: torch.add(small, small, out=o)):.2f} us')

# head backward
lr = torch.randn(2, 4, 65, 65, 65, device=dev)
probs = ops.UpSoftmaxFn.apply(lr, (128, 128, 128), True)
gp = torch.randn_like(probs)
g_lr = torch.empty_like(lr)
nws = L.hno_upsoftmax_bwd_workspace_bytes(2, 4, 65, 65, 65, 128, 128, 128)
ws = torch.empty(nws // 4, device=dev)
P, S = pkg._lib.ptr, pkg._lib.stream_ptr
for dbg, name in ((0, 'full'), (1 << 8, 'nsplit 1'), (2 << 8, 'nsplit 2'), (3 << 8, 'nsplit 3'), (4 << 8, 'nsplit 4'), (8 << 8, 'nsplit 8'), (3, 'neither'), (0x20, 'return at start'), (0x40, 'return after taps'), (0x63, 'neither, no flush'), (0x60, 'no flush')):
    L.hno_set_debug(dbg)
    t = graph_time(lambda: L.hno_upsoftmax_bwd(P(gp), P(probs), P(g_lr), P(ws), 2, 4, 65, 65, 65, 128, 128, 128, 1, 0, S()), n=10)
    print(f'upsoftmax_bwd [{name}]: {t:.1f} us')
L.hno_set_debug(0)
with pkg._lib.KernelProfile() as kp:
    for _ in range(20): L.hno_upsoftmax_bwd(P(gp), P(probs), P(g_lr), P(ws), 2, 4, 65, 65, 65, 128, 128, 128, 1, 0, S())
for k, v in kp.summary().items(): print(k, v)

# conv_in
xin = torch.randn(2, 4, 128, 128, 128, device=dev)
Wk = torch.randn(24, 4, 2, 2, 2, device=dev) * 0.1; bk = torch.randn(24, device=dev) * 0.01
yk = ops.ConvK2S2Fn.apply(xin, Wk, bk, ops.ACT_SELU)
gyk = torch.randn_like(yk); dWk = torch.empty_like(Wk); dbk = torch.empty_like(bk)
wsk = torch.empty(L.hno_pwconv_bwd_workspace_bytes(32, 24) // 4, device=dev)
for grid in (1280, 1536, 2048, 3072):
    L.hno_set_debug(grid << 8)
    tf = graph_time(lambda: L.hno_conv_k2s2_fwd(P(xin), P(Wk), P(bk), P(yk), 2, 4, 24, 128, 128, 128, 1, 0, S()), n=10)
    print(f'conv_k2s2 grid {grid}: fwd {tf:.1f} us')
for grid in (0, 768, 1024):
    L.hno_set_debug(grid << 8)
    tf = graph_time(lambda: L.hno_conv_k2s2_fwd(P(xin), P(Wk), P(bk), P(yk), 2, 4, 24, 128, 128, 128, 1, 0, S()), n=10)
    tb = graph_time(lambda: L.hno_conv_k2s2_bwd(P(gyk), P(yk), P(xin), P(Wk), None, P(dWk), P(dbk), P(wsk), 2, 4, 24, 128, 128, 128, 1, 0, S()), n=10)
    print(f'conv_k2s2 grid {grid or "default"}: fwd {tf:.1f} us, bwd (+reduce) {tb:.1f} us')
L.hno_set_debug(0)

# loss statistics
from multimodal_3d_image_segmentation_amd.nets import custom_losses
pr = torch.softmax(torch.randn(2, 4, 128, 128, 128, device=dev), 1)
lb = torch.randint(0, 4, (2, 128, 128, 128), device=dev, dtype=torch.uint8)
stats = torch.empty(32, device=dev, dtype=torch.float64); coef = torch.empty(2, 4, 4, device=dev); lossv = torch.empty((), device=dev)
for cfg in (16, 128 << 8, 256 << 8, 512 << 8, 1024 << 8, 2048 << 8):
    L.hno_set_debug(cfg)
    t = graph_time(lambda: L.hno_loss_fwd(P(pr), P(lb), P(stats), P(coef), P(lossv), 2, 4, 128 ** 3, 0, 0.0, S()), n=10)
    print(f'loss_fwd cfg {cfg if cfg == 16 else cfg >> 8}: {t:.1f} us (stats + finalize + memset)')
L.hno_set_debug(0)
