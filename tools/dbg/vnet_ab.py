"""V-Net-DS cfg4 bf16 graph-replay step time with a debug flag set (A/B of kernel variants): python tools/dbg/vnet_ab.py 0 4096"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.nets import custom_losses
L = pkg._lib.lib()
dev = 'cuda'
torch.manual_seed(0)
model = pkg.nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4]).to(dev)
x = torch.randn(1, 4, 160, 192, 128, device=dev)
lab = pkg.ops.labels_prepare(torch.randint(0, 4, (1, 1, 160, 192, 128), device=dev).float(), 4)
loss_fn = custom_losses.PCCLoss()
pkg.ops.set_defer_reduce(os.environ.get('HNO_DEFER', '1') == '1')      # batched end-of-backward slab reductions, as bench.py / training() run it
def step():
    for p in model.parameters(): p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        with pkg.ops.expected_loss(lab, loss_fn):
            y = model(x)
        loss = loss_fn(y, lab)
    loss.backward()
for flag in [int(a) for a in sys.argv[1:]] or [0]:
    L.hno_set_debug(flag)
    step(); step(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side): step()
    torch.cuda.current_stream().wait_stream(side)
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f'flag {flag}: {min(ts):.3f} ms per step (graph replay)')
    del gr
L.hno_set_debug(0)
