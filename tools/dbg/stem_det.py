"""is the chained stem's backward deterministic?  python tools/dbg/stem_det.py"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
ops = pkg.ops
for shape in ((2, 2, 16, 20, 24), (2, 4, 128, 128, 128), (2, 4, 32, 32, 32)):
    torch.manual_seed(0)
    Cin = shape[1]
    x = torch.randn(shape, device='cuda')
    C0 = C1 = 8 if Cin == 2 else 24
    ps = [torch.randn(C0, Cin, 2, 2, 2, device='cuda') * .3, torch.randn(C0, device='cuda') * .1, torch.randn(C1, C0, 1, 1, 1, device='cuda') * .2, torch.randn(C1, device='cuda') * .1]
    outs = []
    for it in range(4):
        q = [p.clone().requires_grad_(True) for p in ps]
        y = ops.StemChainFn.apply(x, q[0], q[1], q[2], q[3], ops.ACT_SELU)
        torch.manual_seed(1)
        cot = torch.randn_like(y)
        gs = torch.autograd.grad((y * cot).sum(), q)
        outs.append([y.detach().clone()] + [g.clone() for g in gs])
    for it in range(1, 4):
        print(shape, it, [float((a - b).abs().max()) for a, b in zip(outs[0], outs[it])])
