import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd.nets import custom_losses
from multimodal_3d_image_segmentation_amd.parallel import FlatGradReplica
from multimodal_3d_image_segmentation_amd.experiments.train_test import CapturedStep
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
loss_fn = custom_losses.PCCLoss()
torch.manual_seed(3)
vnet = pkg.nets.VNetDS(2, 3, 8, [1, 1, 1], right_leg_indexes=[0, 1, 2]).cuda()
xv = torch.randn(1, 2, 32, 32, 32, device='cuda')
labv = torch.randint(0, 3, (1, 1, 32, 32, 32)).float()
loss_fn(vnet(xv), ops.labels_prepare(labv.cuda(), 3)).backward()
refv = [p.grad.clone() for p in vnet.parameters()]
for p in vnet.parameters(): p.grad = None
loss_fn(vnet(xv), ops.labels_prepare(labv.cuda(), 3)).backward()
print('eager run-to-run max diff', max(float((p.grad - w).abs().max()) for p, w in zip(vnet.parameters(), refv)))
for p in vnet.parameters(): p.grad = None
repv = FlatGradReplica(vnet, bucket_bytes=64 << 10, min_buckets=3, overlap=True, broadcast=False, force_distributed=True)
print('buckets', [(b[0], b[1], len(b[2])) for b in repv.buckets])
repv.zero_grad()
loss_fn(vnet(xv), ops.labels_prepare(labv.cuda(), 3)).backward()
repv.allreduce_grads(); torch.cuda.synchronize()
print('eager replica max diff', max(float((p.grad - w).abs().max()) for p, w in zip(vnet.parameters(), refv)))
os.environ['HNO_DP_CAPTURE_ALLREDUCE'] = '1'
for bucketed in (False, True):
    capv = CapturedStep(vnet, loss_fn, 3, None, repv, bucketed=bucketed)
    print('bucketed', capv.bucketed, 'first', capv.step(xv, labv))
    for i in range(2):
        repv.flat_grad.fill_(77.0)
        l = capv.step(xv, labv)
        torch.cuda.synchronize()
        bad = [(n, float((p.grad - w).abs().max()), float(w.abs().max())) for (n, p), w in zip(vnet.named_parameters(), refv)
               if float((p.grad - w).abs().max()) > 1e-5 * float(w.abs().max()) + 1e-12]
        print(' replay', i, 'loss', None if l is None else float(l), 'mismatching', len(bad), bad[:6])
repv.close()
dist.destroy_process_group()
