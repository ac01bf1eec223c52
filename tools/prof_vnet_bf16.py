"""Per-call breakdown of one V-Net-DS cfg4 step on the bf16 path: every hno kernel launch with its HIP-event time and
algorithmic flops (grouped by kernel and problem size)."""
import sys, os, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.nets import custom_losses
torch.manual_seed(0)
model = pkg.nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4]).cuda()
x = torch.randn((1, 4, 160, 192, 128), device='cuda')
lab = pkg.ops.labels_prepare(torch.randint(0, 4, (1, 1, 160, 192, 128), device='cuda').float(), 4)
def step():
    for p in model.parameters(): p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = custom_losses.PCCLoss()(model(x), lab)
    loss.backward()
step(); step()
with pkg._lib.KernelProfile() as kp:
    step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, ms, nb in kp.records:
    k = (name, round(nb / 1e9, 3))
    c, t = agg.get(k, (0, 0.0))
    agg[k] = (c + 1, t + ms)
tot = sum(t for _, t in agg.values())
print(f'total profiled kernel time {tot:.2f} ms')
for (name, gf), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{name:28s} {gf:9.3f} G(flop|B)  x{c:3d}  {t:7.3f} ms  {gf * c / t if t else 0:8.1f} T/s')
