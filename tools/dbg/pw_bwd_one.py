import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
L = pkg._lib.lib(); P, S = pkg._lib.ptr, pkg._lib.stream_ptr
dev = 'cuda'; B, C, N = 2, 24, int(os.environ.get('PW_N', '65'))
xa = torch.randn(B, C, N, N, N, device=dev); xb = torch.randn_like(xa)
W = torch.randn(C, 2 * C, device=dev) * 0.1; bias = torch.randn(C, device=dev) * 0.01
y = ops.PwConvFn.apply(xa, xb, W, bias, ops.ACT_SELU)
gy = torch.randn_like(y); gxa, gxb = torch.empty_like(xa), torch.empty_like(xb)
dW, db = torch.empty_like(W), torch.empty_like(bias)
ws = torch.empty(L.hno_pwconv_bwd_workspace_bytes(48, 24) // 4, device=dev)
junk = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
for flags in [int(a) for a in sys.argv[1:]] or [0]:
    L.hno_set_debug(flags)
    for _ in range(5):
        junk.fill_(1)      # flush the caches between launches
        L.hno_pwconv_bwd(P(gy), P(y), P(xa), 24, P(xb), 24, P(W), P(gxa), P(gxb), P(dW), P(db), P(ws), B, 24, N ** 3, 1, 0, 0, S())
    torch.cuda.synchronize()
L.hno_set_debug(0)
