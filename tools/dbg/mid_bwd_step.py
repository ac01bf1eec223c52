"""Along ONE Adamax trajectory (steps taken with the three-kernel backward), the gradient of the fused backward middle against the
three-kernel one at every step: where do they part?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops, optim
from multimodal_3d_image_segmentation_amd.nets.hnosegxs import HNOSegXS

torch.manual_seed(0)
img = torch.randn(2, 4, 128, 128, 128, device='cuda')
lab = torch.randint(0, 4, (2, 1, 128, 128, 128), device='cuda').to(torch.uint8)
torch.manual_seed(1)
net = HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14), device='cuda')
opt = optim.Adamax(net.parameters(), lr=1e-2)
names = [n for n, _ in net.named_parameters()]
for step in range(8):
    gr = {}
    for flag in ('1', '0'):
        os.environ['HNO_FUSED_MID_BWD'] = flag
        opt.zero_grad(set_to_none=True)
        loss, _ = ops.SegLossFn.apply(net(img), lab, 0, 0.0)
        loss.backward()
        gr[flag] = [p.grad.clone() for p in net.parameters()]
    worst = max(((a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30), n, b.abs().max().item(),
                 bool(torch.isfinite(a).all()))
                for a, b, n in zip(gr['1'], gr['0'], names))
    zmax = max(float(p.abs().max()) for p in net.parameters())
    print(f'step {step}: loss {float(loss.detach()):.6f}  worst rel diff {worst[0]:.3e} at {worst[1]} (max |g| {worst[2]:.3e}, finite {worst[3]}); max |param| {zmax:.3f}')
    opt.step()
