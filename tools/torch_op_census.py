"""Which ATen ops (= kernels that are not ours) are still launched in one HNOSeg-XS training step."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.nets import custom_losses
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
model = pkg.nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14)).cuda()
opt = torch.optim.Adamax(model.parameters(), lr=5e-3)
x = torch.randn(2, 4, 128, 128, 128, device='cuda')
lab = pkg.ops.labels_prepare(torch.randint(0, 4, (2, 1, 128, 128, 128), device='cuda').float(), 4)
loss_fn = custom_losses.PCCLoss()
def step():
    y = model(x); loss = loss_fn(y, lab)
    for p in model.parameters(): p.grad = None
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') and e.key.split('::')[1] in
        ('fill_', 'zero_', 'zeros', 'zeros_like', 'ones_like', 'copy_', 'add', 'add_', 'cat', 'stack', 'clone', 'contiguous', 'mul', 'sum', 'empty_like', 'full', 'full_like', '_to_copy', 'to')]
for e in sorted(rows, key=lambda e: -e.count)[:25]:
    print(f'{e.key:22s} x{e.count:3d}  shapes {str(e.input_shapes)[:110]}')
