"""Data-parallel replicas: one process per GPU, one flat fp32 gradient buffer, one RCCL
all-reduce per step over xGMI (``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is
used by the CPU tests).

The reference has no distributed code; volumes are independent, so the
only exchange step is the gradient mean.  Backward leaves every gradient where its kernel wrote
it (``.grad`` is reset to None, so autograd installs the fresh tensors without an accumulation
kernel per parameter); before the collective they are packed into one flat buffer by a single
multi-tensor copy, and afterwards every ``.grad`` is a VIEW into that buffer.  HNOSeg-XS is a
single 113 KB message (latency-bound); large models are split into buckets.
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
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += p.numel()
        per = max(1, bucket_bytes // 4)
        self.buckets = [self.flat_grad[i:i + per] for i in range(0, n, per)]
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._avg = self.world > 1 and dist.get_backend(process_group) == 'nccl'   # RCCL reduces with AVG directly
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
        for p in self.params:   # no kernel: the next backward installs fresh gradient tensors
            p.grad = None

    def pack(self, grads=None):
        """Copies the gradients (default: the current ``.grad`` of every parameter) into the flat buffer
        with one multi-tensor kernel and makes ``.grad`` the views.  Gradients that already are the
        views (no zero_grad since the last step: autograd accumulated into them) are left alone."""
        grads = [p.grad for p in self.params] if grads is None else grads
        src, dst = [], []
        for g, v in zip(grads, self.views):
            if g is None:
                v.zero_()
            elif g is not v:
                src.append(g)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def allreduce_grads(self, grads=None, async_op=False):
        """Mean of the gradients over replicas == the single-process gradient of the mean loss over
        the concatenated batch (PCCLoss is a mean over (b, c) with equal per-rank batch).
        `grads`: the tensors backward wrote (needed when backward is replayed from a HIP graph and
        ``.grad`` no longer names them); default the current ``.grad``s.  With one replica nothing is
        copied or sent."""
        if self.world == 1:
            return None
        self.pack(grads)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        works = [dist.all_reduce(b, op=op, group=self.group, async_op=True) for b in self.buckets]
        if async_op:
            return works
        for w in works:
            w.wait()
        if not self._avg:
            self.flat_grad.mul_(1.0 / self.world)
        return None

    def __call__(self, *a, **k):
        return self.module(*a, **k)
