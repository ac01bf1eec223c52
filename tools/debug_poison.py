"""Poison every buffer ops.py allocates with NaN: an output element a kernel fails to write shows up as NaN downstream."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd.nets import custom_losses
from _inputs import formula_volume, formula_labels, NOSEG_MODELS

_empty, _empty_like = torch.empty, torch.empty_like
class _T:
    def __getattr__(self, k): return getattr(torch, k)
    @staticmethod
    def empty(*a, **k):
        t = _empty(*a, **k)
        return t.fill_(float('nan')) if t.is_floating_point() else t
    @staticmethod
    def empty_like(*a, **k):
        t = _empty_like(*a, **k)
        return t.fill_(float('nan')) if t.is_floating_point() else t
ops.torch = _T()

name = sys.argv[1] if len(sys.argv) > 1 else 'hnoseg_deep_supervision'
g = np.load(os.path.join(ROOT, 'tests/golden/g7_noseg_models.npz'))
kw, shape = NOSEG_MODELS[name]
model = pkg.nets.NeuralOperatorSeg(**kw)
pre = f'{name}::sd::'
model.load_state_dict({k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)})
model = model.cuda()
K = kw['out_channels']
x = torch.from_numpy(formula_volume(shape, 4)).cuda()
lab = torch.from_numpy(formula_labels((shape[0], 1) + shape[2:], K, 6)).cuda()
y = model(x)
print('nan in y', int(torch.isnan(y).sum()))
loss = custom_losses.PCCLoss()(y, pkg.ops.labels_prepare(lab, K))
loss.backward()
for k, p in model.named_parameters():
    print(f'{k:40s} nan={int(torch.isnan(p.grad).sum())} of {p.grad.numel()}')
