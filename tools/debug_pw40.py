import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
torch.manual_seed(0)
for C, Co in ((40, 40), (24, 40), (40, 24), (64, 48)):
    x = torch.randn(2, C, 3, 5, 7, device='cuda')
    W = torch.randn(Co, C, device='cuda') * 0.3
    y = ops.pwconv_fwd_raw(x, None, W, None, ops.ACT_NONE)
    ref = torch.einsum('oi,bidhw->bodhw', W.double(), x.double())
    err = (y.double() - ref).abs().amax(dim=(0, 2, 3, 4))
    print(C, Co, 'max err per out channel:', [f'{e:.1e}' for e in err.tolist()])
