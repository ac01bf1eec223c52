import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
torch.manual_seed(0)
for B, V in ((1, 32768), (2, 32768), (1, 4096), (2, 274625), (1, 35937), (3, 1000), (1, 33)):
    for Ca, Cb, Co in ((24, 24, 24), (24, 0, 24), (24, 0, 4), (48, 0, 48)):
        xa = torch.randn(B, Ca, V, 1, 1, device='cuda'); xb = torch.randn(B, Cb, V, 1, 1, device='cuda') if Cb else None
        W = torch.randn(Co, Ca + Cb, device='cuda') * 0.2; bias = torch.randn(Co, device='cuda') * 0.1
        for act in (ops.ACT_SELU, ops.ACT_NONE):
            y = ops.PwConvFn.apply(xa, xb, W, bias, act)
            xin = xa if xb is None else torch.cat([xa, xb], 1)
            ref = torch.einsum('oi,bivxy->bovxy', W.double(), xin.double()) + bias.double().view(1, -1, 1, 1, 1)
            if act == ops.ACT_SELU: ref = torch.nn.functional.selu(ref)
            err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
            print(f'B {B} V {V} {Ca}+{Cb}->{Co} act {act}: rel err {err:.2e}', 'BAD' if err > 1e-5 else '')
