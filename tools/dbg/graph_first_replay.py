"""Probe: an eager kernel launch between a HIP-graph capture of a training step and the FIRST replay leaves that replay's loss NaN on
this stack (DESIGN lesson 36).  python tools/dbg/graph_first_replay.py <repo root> <mode: plain | other | fill | zero | norep_other ...>"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import multimodal_3d_image_segmentation_amd as pkg
from multimodal_3d_image_segmentation_amd import ops
from multimodal_3d_image_segmentation_amd.nets import custom_losses
from multimodal_3d_image_segmentation_amd.parallel import FlatGradReplica
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
torch.manual_seed(0)
model = pkg.nets.HNOSegXS(2, 3, 8, [1, 1, 1, 1], (3, 3, 3)).cuda()
x = torch.randn(2, 2, 16, 16, 16, device='cuda')
lab = ops.labels_prepare(torch.randint(0, 3, (2, 1, 16, 16, 16), device='cuda').float(), 3)
loss_fn = custom_losses.PCCLoss()
loss_fn(model(x), lab).backward()
ref = [p.grad.clone() for p in model.parameters()]
for p in model.parameters(): p.grad = None
mode = sys.argv[2]
rep = FlatGradReplica(model, min_buckets=3, overlap=True, broadcast=False, force_distributed='norep' not in mode)
rep.set_hooks_enabled(False)
ops.set_defer_reduce('defer' in mode)
def fwd_bwd():
    rep.zero_grad()
    y = model(x)
    l = loss_fn(y, lab)
    l.backward()
    return y, l
fwd_bwd(); torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
        y, l = fwd_bwd()
        rep.finish_capture()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
other = torch.empty(1000, device='cuda')
for replay in range(3):
    if 'fill' in mode: rep.flat_grad.fill_(123.0)
    if 'other' in mode: other.fill_(5.0)
    if 'sync' in mode: torch.cuda.synchronize()
    if 'zero' in mode: rep.flat_grad.zero_()
    graph.replay(); torch.cuda.synchronize()
    bad = [n for (n, p), w in zip(model.named_parameters(), ref) if not torch.equal(p.grad, w)]
    nanp = [n for (n, p) in model.named_parameters() if bool(torch.isnan(p.grad).any())]
    print(f'{mode} replay {replay} before allreduce: y nan {int(torch.isnan(y).sum())}, loss {float(l):.6f}, mismatching {len(bad)}, nan {len(nanp)} {nanp[:2]}')
    if 'ar' in mode:
        rep.allreduce_flat(); torch.cuda.synchronize()
        bad = [n for (n, p), w in zip(model.named_parameters(), ref) if not torch.equal(p.grad, w)]
        print(f'   after allreduce: mismatching {len(bad)}')
rep.close()
dist.destroy_process_group()
