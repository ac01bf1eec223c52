"""which weight-gradient reductions of a HartleyMHASeg backward are deferred, which are not (and who asked)?"""
import sys, os, torch, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd.nets import custom_losses
ops.set_defer_reduce(True)
cnt = collections.Counter()
orig = ops._DeferReduce.__init__
def init(self, ok=False):
    fr = [f for f in traceback.extract_stack()[:-1] if 'ops' in f.filename]
    who = ' <- '.join(f.name for f in fr[-3:])
    cnt[(bool(ok), who)] += 1
    orig(self, ok)
ops._DeferReduce.__init__ = init
shapes = collections.Counter()
orig_bwd = ops.pwconv_bwd_raw
def bwd(gy, y, xa, xb, W, act, has_bias, *a, **k):
    shapes[(tuple(W.shape), bool(k.get('defer', False)), W.is_leaf, None if k.get('bias') is None else k['bias'].is_leaf)] += 1
    return orig_bwd(gy, y, xa, xb, W, act, has_bias, *a, **k)
ops.pwconv_bwd_raw = bwd
torch.manual_seed(0)
model = pkg.nets.HartleyMHASeg(4, 4, 12, 16, 4, (10, 14, 14), (2, 2, 2)).cuda()
x = torch.randn(1, 4, 128, 128, 128, device='cuda')
lab = ops.labels_prepare(torch.randint(0, 4, (1, 1, 128, 128, 128), device='cuda').float(), 4)
for it in range(2):
    for p in model.parameters(): p.grad = None
    cnt.clear(); shapes.clear()
    loss = custom_losses.PCCLoss()(model(x), lab); loss.backward()
for k, n in sorted(cnt.items(), key=lambda kv: -kv[1]): print(n, k)

for k, n in sorted(shapes.items(), key=lambda kv: -kv[1]): print(n, k)
