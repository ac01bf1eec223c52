import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib(); P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev = 'cuda'
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
B, Cin, Cout, N = 2, 4, 24, 128
x = torch.randn(B, Cin, N, N, N, device=dev); W = torch.randn(Cout, Cin, 2, 2, 2, device=dev) * 0.2; bias = torch.randn(Cout, device=dev) * 0.1
No = N // 2 + 1
y = torch.empty(B, Cout, No, No, No, device=dev)
ref = torch.nn.functional.selu(torch.nn.functional.conv3d(x, W, bias, stride=2, padding=1))
for grid in (0, 256, 512, 768, 1024):
    L.hno_set_debug(grid << 8)
    t = timeit(lambda: L.hno_conv_k2s2_fwd(P(x), P(W), P(bias), P(y), B, Cin, Cout, N, N, N, 1, 0, S()))
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    print(f'k2s2 fwd grid {grid or "default"}: {t:.1f} us, rel err {err:.2e}')
L.hno_set_debug(0)
gy = torch.randn_like(y); dW = torch.empty_like(W); db = torch.empty_like(bias)
ws = torch.empty(L.hno_conv_k2s2_bwd_workspace_bytes(Cin, Cout) // 4, device=dev) if hasattr(L, 'hno_conv_k2s2_bwd_workspace_bytes') else torch.empty(1 << 22, device=dev)
for grid in (0, 256, 512, 1024):
    L.hno_set_debug(grid << 8)
    t = timeit(lambda: L.hno_conv_k2s2_bwd(P(gy), P(y), P(x), P(W), None, P(dW), P(db), P(ws), B, Cin, Cout, N, N, N, 1, 0, S()))
    print(f'k2s2 bwd grid {grid or "default"}: {t:.1f} us (incl. slab reduce)')
L.hno_set_debug(0)
xr = x.clone().requires_grad_(False); Wr = W.clone().requires_grad_(True); br = bias.clone().requires_grad_(True)
yr = torch.nn.functional.selu(torch.nn.functional.conv3d(x, Wr, br, stride=2, padding=1)); yr.backward(gy)
L.hno_conv_k2s2_bwd(P(gy), P(y), P(x), P(W), None, P(dW), P(db), P(ws), B, Cin, Cout, N, N, N, 1, 0, S())
print('dW rel err', ((dW - Wr.grad).abs().max() / Wr.grad.abs().max()).item(), 'db', ((db - br.grad).abs().max() / br.grad.abs().max()).item())
