"""HNOSeg-XS step at 2 x 4 x 96^3 (49^3 working grid: planes the generic kernels serve) -- graph-replayed fwd + PCC + bwd in ms.
HNO_GENERIC_WAVES_MID selects the waves per plane of the generic plane kernels (4 / 8 / 16).  python tools/dbg/midplane_ab.py [size]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.nets import custom_losses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
torch.manual_seed(0)
model = pkg.nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14)).cuda()
x = torch.randn(2, 4, n, n, n, device='cuda')
lab = torch.randint(0, 4, (2, n, n, n), device='cuda').to(torch.uint8)
loss_fn = custom_losses.PCCLoss()


def step():
    loss = loss_fn(model(x), lab)
    for p in model.parameters():
        p.grad = None
    pkg.ops.backward_from(loss)
    return loss


for _ in range(3):
    l = step()
del l
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        sl = step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f'{n}^3: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step (waves per mid plane: {os.environ.get("HNO_GENERIC_WAVES_MID", "4")})')
