"""A/B of the fused backward spectral middle at the benchmark configuration: every parameter gradient with HNO_FUSED_MID_BWD=1 vs 0."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd.nets.hnosegxs import HNOSegXS

torch.manual_seed(0)
img = torch.randn(2, 4, 128, 128, 128, device='cuda')
lab = torch.randint(0, 4, (2, 1, 128, 128, 128), device='cuda').to(torch.uint8)
res = {}
for flag in ('0', '1'):
    os.environ['HNO_FUSED_MID_BWD'] = flag
    torch.manual_seed(1)
    net = HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14), device='cuda')
    if len(sys.argv) > 2:    # weights off their initial values (biases non-zero, mixes perturbed)
        with torch.no_grad():
            for p in net.parameters():
                p.add_(float(sys.argv[2]) * torch.randn_like(p))
    for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
        for p in net.parameters():
            p.grad = None
        loss, _ = ops.SegLossFn.apply(net(img), lab, 0, 0.0)
        loss.backward()
    res[flag] = (float(loss), {n: p.grad.clone() for n, p in net.named_parameters()})
print('loss', res['0'][0], res['1'][0])
worst = []
for n in res['0'][1]:
    a, b = res['0'][1][n].double(), res['1'][1][n].double()
    worst.append(((a - b).abs().max().item() / max(a.abs().max().item(), 1e-30), a.abs().max().item(), n))
for w in sorted(worst, reverse=True)[:12]:
    print(f'{w[2]:50s} rel diff {w[0]:.3e}   max |g| {w[1]:.3e}')
