"""Experiment: two independent batch-1 HNOSeg-XS training steps on two streams inside one HIP graph vs one batch-2 step.
python tools/dbg/two_stream.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd.nets import custom_losses
import bench
dev = 'cuda'
loss_fn = custom_losses.PCCLoss()
pkg.ops.set_defer_reduce(True)


def make(B, seed):
    torch.manual_seed(seed)
    m = pkg.nets.HNOSegXS(**bench.MODEL_CFG).to(dev)
    x = torch.randn((B, 4) + bench.VOL, device=dev)
    lab = pkg.ops.labels_prepare(torch.randint(0, 4, (B, 1) + bench.VOL, device=dev).float(), 4)
    return m, x, lab


def step(m, x, lab):
    for p in m.parameters():
        p.grad = None
    with pkg.ops.expected_loss(lab, loss_fn):
        y = m(x)
    loss = loss_fn(y, lab)
    pkg.ops.backward_from(loss)
    return loss


def timeit(graph, n=30):
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def capture(fn):
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn(side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return g


m2, x2, l2 = make(2, 0)
step(m2, x2, l2); step(m2, x2, l2)
g2 = capture(lambda side: step(m2, x2, l2))
print('one stream, batch 2: %.3f ms' % timeit(g2))

ma, xa, la = make(1, 1)
mb, xb, lb = make(1, 2)
step(ma, xa, la); step(mb, xb, lb); step(ma, xa, la); step(mb, xb, lb)
g1 = capture(lambda side: (step(ma, xa, la), step(mb, xb, lb)))
print('one stream, 2 x batch 1 back to back: %.3f ms' % timeit(g1))


def two(side):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(side); s2.wait_stream(side)
    with torch.cuda.stream(s1):
        step(ma, xa, la)
    with torch.cuda.stream(s2):
        step(mb, xb, lb)
    side.wait_stream(s1); side.wait_stream(s2)


gt = capture(two)
print('two streams, 2 x batch 1: %.3f ms' % timeit(gt))
