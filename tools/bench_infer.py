"""Inference benchmark of the reference's only PUBLISHED numbers (README.md:10, docs/assets/images/computational_efficiency.png:
HNOSeg-XS single-image inference < 0.24 s and < 1.8 GiB on a Tesla V100, images 240 x 240 x 155).

Protocol of experiments/train_test.py:383-426: batch 1, model.eval(), no_grad, the timed span covers the host->device copy
of x, the forward pass and the copy of the prediction back to the host; the first sample is excluded; peak memory from
torch.cuda.max_memory_{reserved,allocated}.  Here the class map is produced on the GPU (argmax fused into the upsampling
kernel), so one uint8 volume crosses PCIe instead of 4 float32 probability volumes.

    python tools/bench_infer.py [--size 240 240 155] [--samples 8] [--model hnosegxs|fnoseg|hnoseg|vnetds]
prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
nets = pkg.nets
MODELS = {
    'hnosegxs': lambda: nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14)),                       # config_hnoseg_xs.ini
    'fnoseg': lambda: nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier'),          # config_fnoseg.ini
    'hnoseg': lambda: nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Hartley'),          # config_hnoseg.ini
    'vnetds': lambda: nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4]),
}
ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, nargs=3, default=[240, 240, 155])
ap.add_argument('--samples', type=int, default=8)
ap.add_argument('--model', default='hnosegxs', choices=list(MODELS))
args = ap.parse_args()
torch.manual_seed(0)
model = MODELS[args.model]().cuda().eval()
size = tuple(args.size)
g = torch.Generator().manual_seed(1)
xs = [torch.randn((1, 4) + size, generator=g).pin_memory() for _ in range(2)]
torch.cuda.reset_peak_memory_stats()
times, gpu_times = [], []
for i in range(args.samples + 1):
    x_host = xs[i % 2]
    torch.cuda.synchronize()
    t0 = time.time()
    x = x_host.to('cuda', non_blocking=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad(), ops.label_output():
        yp = model(x)
    e1.record()
    y_pred = np.asarray(yp.to('cpu'))[0, 0]
    t1 = time.time()
    if i:          # the first sample includes kernel-table set-up (the reference skips it too, train_test.py:415-416)
        times.append(t1 - t0)
        gpu_times.append(e0.elapsed_time(e1) * 1e-3)
assert y_pred.shape == size and y_pred.dtype == np.uint8
print(json.dumps({
    'metric': 'single-image inference time, protocol of experiments/train_test.py:383-426', 'model': args.model, 'image_size': size,
    'seconds_per_image': round(float(np.mean(times)), 5), 'gpu_forward_seconds': round(float(np.mean(gpu_times)), 5),
    'max_memory_reserved_MiB': round(torch.cuda.max_memory_reserved() / 1024 ** 2, 1),
    'max_memory_allocated_MiB': round(torch.cuda.max_memory_allocated() / 1024 ** 2, 1),
    'samples': args.samples, 'dtype': 'f32', 'data': 'synthetic',
    'reference_published': 'HNOSeg-XS < 0.24 s, < 1.8 GiB on a Tesla V100 (README.md:10; ~0.20 s / ~1.55 GiB in Fig. 1)',
}))
