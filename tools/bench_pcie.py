"""PCIe-inclusive rate of the HNOSeg-XS training step: the batch (2 x 4 x 128^3 fp32 + labels) starts in HOST memory every step, pinned
or pageable, and goes through experiments.train_test.CapturedStep (host batch copied straight into the captured step's input buffers,
then the graph replay, then Adamax).  bench.py's `value` is measured with the inputs resident in HBM; this is the other number DESIGN.md
section 5 quotes.   python tools/bench_pcie.py"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.nets import custom_losses
from multimodal_3d_image_segmentation_amd.experiments.train_test import CapturedStep

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = pkg.nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14)).to(dev)
opt = pkg.optim.Adamax(model.parameters(), lr=5e-3)
cap = CapturedStep(model, custom_losses.PCCLoss(), 4)
B = 2
xh = [torch.randn(B, 4, 128, 128, 128) for _ in range(2)]
yh = [torch.randint(0, 4, (B, 1, 128, 128, 128)).float() for _ in range(2)]
out = {}
for tag, pin in (('pageable', False), ('pinned', True)):
    xs = [t.pin_memory() if pin else t for t in xh]
    ys = [t.pin_memory() if pin else t for t in yh]

    def step(i):
        loss = cap.step(xs[i & 1], ys[i & 1])
        if loss is None:      # first occurrence of the shape: eager
            loss = custom_losses.PCCLoss()(model(xs[i & 1].to(dev)), pkg.experiments.utils.labels_to_u8(ys[i & 1].to(dev), 4, None))
            opt.zero_grad()
            loss.backward()
            loss = None
        opt.step()
    for overlap in (False, True):
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
            if overlap:       # what training() does: the next batch crosses PCIe on the copy stream while this step runs
                cap.prefetch(xs[(i + 1) & 1], ys[(i + 1) & 1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[tag + ('_prefetched' if overlap else '_serial')] = {'ms_per_step': round(dt * 1e3, 3), 'volumes_per_s': round(B / dt, 1)}
print(json.dumps({'metric': 'HNOSeg-XS training step with the batch in host memory (PCIe-inclusive)', 'batch': B, **out}))
