"""ATen ops (kernels that are not ours) in one bf16 V-Net-DS cfg4 step, with the Python line that issued them."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.nets import custom_losses
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
model = pkg.nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4]).cuda()
x = torch.randn(1, 4, 160, 192, 128, device='cuda')
lab = pkg.ops.labels_prepare(torch.randint(0, 4, (1, 1, 160, 192, 128), device='cuda').float(), 4)
loss_fn = custom_losses.PCCLoss()
def step():
    for p in model.parameters(): p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = loss_fn(model(x), lab)
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
keep = ('fill_', 'zero_', 'zeros', 'zeros_like', 'copy_', 'add', 'add_', 'cat', 'clone', 'contiguous', 'mul', 'sum', '_to_copy', 'to', 'slice', 'index', 'select')
import collections
agg = collections.Counter()
for e in prof.events():
    if e.name.startswith('aten::') and e.name.split('::')[1] in ('fill_', 'zero_', 'copy_', 'add', 'add_', 'cat', 'mul', 'sum', 'clone'):
        st = [s for s in (e.stack or []) if 'multimodal' in s or 'tools/' in s]
        agg[(e.name, str(e.input_shapes)[:70], st[0][-70:] if st else '?')] += 1
for (k, n) in agg.most_common(60):
    print(f'x{n:3d} {k[0]:14s} {k[1]:72s} {k[2]}')
