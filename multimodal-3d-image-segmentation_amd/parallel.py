"""Data-parallel replicas: one process per GPU, one flat fp32 gradient buffer, one RCCL
all-reduce per step over xGMI (``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is
used by the CPU tests).

The reference has no distributed code; volumes are independent, so the
only exchange step is the gradient mean.  Every parameter's ``.grad`` is a VIEW into one flat
buffer, so autograd accumulates straight into it and the collective needs no packing copies:
HNOSeg-XS is a single 113 KB message (latency-bound), large models are split into buckets.
"""
import torch
import torch.distributed as dist


class FlatGradReplica:
    """Wraps a module for data-parallel training with a flat gradient buffer.

    usage per step:  rep.zero_grad(); loss = ...; loss.backward(); rep.allreduce_grads(); opt.step()
    """

    def __init__(self, module, process_group=None, bucket_bytes=64 << 20, broadcast=True):
        self.module = module
        self.group = process_group
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        per = max(1, bucket_bytes // 4)
        self.buckets = [self.flat_grad[i:i + per] for i in range(0, n, per)]
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if broadcast and self.world > 1:
            self.broadcast_parameters()

    def broadcast_parameters(self, src=0):
        """Identical initial weights on every replica (rank `src` wins)."""
        flat = torch.cat([p.detach().reshape(-1) for p in self.module.parameters()])
        dist.broadcast(flat, src, group=self.group)
        off = 0
        with torch.no_grad():
            for p in self.module.parameters():
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def zero_grad(self):
        self.flat_grad.zero_()   # one memset; keeps the .grad views alive

    def allreduce_grads(self, async_op=False):
        """Mean of the gradients over replicas == the single-process gradient of the mean loss over
        the concatenated batch (PCCLoss is a mean over (b, c) with equal per-rank batch)."""
        if self.world == 1:
            return None
        works = []
        for b in self.buckets:
            works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if async_op:
            return works
        for w in works:
            w.wait()
        self.flat_grad.mul_(1.0 / self.world)
        return None

    def __call__(self, *a, **k):
        return self.module(*a, **k)
