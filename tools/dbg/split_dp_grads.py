"""which gradients need settling in the data-parallel captured step with the sample split?  python tools/dbg/split_dp_grads.py"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd.nets import custom_losses
from multimodal_3d_image_segmentation_amd.parallel import FlatGradReplica
from multimodal_3d_image_segmentation_amd.experiments.train_test import SampleSplit
import bench
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
torch.manual_seed(0)
model = pkg.nets.HNOSegXS(**bench.MODEL_CFG).cuda()
rep = FlatGradReplica(model, force_distributed=True)
loss_fn = custom_losses.PCCLoss()
x = torch.randn((2, 4) + bench.VOL, device='cuda')
lab = ops.labels_prepare(torch.randint(0, 4, (2, 1) + bench.VOL, device='cuda').float(), 4)
for _ in range(2):
    rep.zero_grad()
    with ops.expected_loss(lab, loss_fn):
        y = model(x)
    loss_fn(y, lab).backward()
    del y
rep.set_hooks_enabled(False)
split = SampleSplit(model)
with torch.no_grad():
    model(x[:1])
torch.cuda.synchronize()
names = [n for n, p in model.named_parameters() if p.requires_grad]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
prev = ops.set_defer_reduce(True)
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
        split.fwd_bwd(x, lab, loss_fn, zero_grad=rep.zero_grad)
        lo, hi = rep.flat_grad.data_ptr(), rep.flat_grad.data_ptr() + 4 * rep.flat_grad.numel()
        for n, p, q, v in zip(names, rep.params, split.tparams, rep.views):
            st = 'None' if p.grad is None else ('in place' if p.grad.data_ptr() == v.data_ptr() else ('other slice' if lo <= p.grad.data_ptr() < hi else 'elsewhere'))
            print('%-44s model: %-11s twin: %s' % (n, st, 'None' if q.grad is None else 'tensor'))
        rep.finish_capture()
ops.set_defer_reduce(prev)
dist.destroy_process_group()
