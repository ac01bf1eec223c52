"""Data-parallel replicas: one process per GPU, one flat fp32 gradient buffer cut into buckets, RCCL all-reduce of each
bucket over xGMI as soon as backward has produced it (``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is used by
the CPU tests).

The reference has no distributed code; volumes are independent, so the only exchange step is the
gradient mean.  Design:

* ``flat_grad`` holds every parameter's gradient in registration order; ``views[i]`` is parameter i's slice.  The views are
  registered with ``ops.set_grad_destinations`` so the backward kernels WRITE their weight gradients there (no pack copy) and
  autograd installs the view as ``.grad``.  Gradients that arrive elsewhere (ops without destination support, accumulated
  gradients) are copied into their view when they are ready.
* Buckets are contiguous ranges of the flat buffer, filled from the END of the parameter list (backward produces the last
  layers' gradients first).  A post-accumulate-grad hook per parameter counts a bucket down; when it reaches zero the bucket
  is all-reduced asynchronously on the communication stream, behind an event recorded on the compute stream -- the collective
  runs while the earlier layers' backward is still computing.  ``allreduce_grads()`` after ``backward()`` launches whatever
  is left (parameters without gradient count as zero) and makes the compute stream wait for the collectives.
* xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): a ring collective is bound by ONE link's bandwidth for large
  messages and by latency (~10-20 us) for small ones.  HNOSeg-XS has 113 KB of gradients: two buckets (the first one hidden
  under the second half of backward, the last one ~15 us exposed); V-Net-DS has 90 MB: 8 MB buckets, ~0.1 ms each at ~77 GB/s
  per direction and link with the ring over all 8 GPUs, all but the last hidden under a 10 ms backward.
"""
import torch
import torch.distributed as dist


class FlatGradReplica:
    """Wraps a module for data-parallel training.

    usage per step:  rep.zero_grad(); loss = ...; loss.backward(); rep.allreduce_grads(); opt.step()
    """

    def __init__(self, module, process_group=None, bucket_bytes=8 << 20, broadcast=True, overlap=True, min_buckets=2,
                 force_distributed=False):
        """force_distributed: take the world > 1 code paths on a group of ONE rank (all-reduce with AVG over one rank is the
        identity): how the data-parallel step is exercised and timed on a single-GPU box (bench.py `dp_path_1rank_ms_per_step`,
        tests)."""
        self.module = module
        self.group = process_group
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.device = dev
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.views, self.offsets, off = [], [], 0
        for p in self.params:
            self.views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            self.offsets.append(off)
            off += p.numel()
        self.real_world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # `world` selects the code paths (> 1: collectives, hooks, flat-buffer destinations); the averaging divisor is always the
        # REAL group size (a forced one-rank group on a backend without AVG halved every gradient)
        self.world = self.real_world
        if force_distributed and dist.is_initialized() and self.world == 1:
            self.world = 2       # branch selector only: the collectives still run over the real (one-rank) group
        self.forced = bool(force_distributed)
        self._avg = self.world > 1 and dist.get_backend(process_group) == 'nccl'   # RCCL reduces with AVG directly
        self._flat_ready = False
        self.overlap = bool(overlap) and self.world > 1
        # ---- buckets: contiguous parameter ranges, cut from the end; at least `min_buckets` so that even a 113 KB model
        # sends its first half while the second half of backward still runs
        per = max(1, min(bucket_bytes // 4, -(-n // max(1, min_buckets))))
        self.bucket_of = [0] * len(self.params)
        self.buckets = []          # (start element, end element, [parameter indices]) in LAUNCH order (last layers first)
        hi, members, size = n, [], 0
        for i in reversed(range(len(self.params))):
            members.append(i)
            size += self.params[i].numel()
            if size >= per or i == 0:
                lo = self.offsets[i]
                self.buckets.append((lo, hi, list(members)))
                for m in members:
                    self.bucket_of[m] = len(self.buckets) - 1
                hi, members, size = lo, [], 0
        self._pending = [len(b[2]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._comm_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        self._hooks = []
        if self.world > 1:
            if broadcast and not self.forced:
                self.broadcast_parameters()
            from . import ops
            if dev.type == 'cuda':
                ops.set_grad_destinations(dict(zip(self.params, self.views)))
                # a deferred (end-of-backward) reduction would land after the bucket was sent; close() restores the setting
                self._prev_defer = ops.set_defer_reduce(False)
            if self.overlap:
                for i, p in enumerate(self.params):
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    # ------------------------------------------------------------------------------------------------------------
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
        for p in self.params:   # no kernel: the next backward installs fresh gradient tensors (or the registered views)
            p.grad = None
        self._pending = [len(b[2]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        if self.world > 1 and self.device.type == 'cuda':
            from . import ops
            ops.new_grad_pass()

    def _settle(self, i, grad=None):
        """gradient of parameter i -> its view (no-op when the kernel already wrote it there)"""
        p, v = self.params[i], self.views[i]
        g = p.grad if grad is None else grad
        if g is None:
            v.zero_()
        elif g.data_ptr() != v.data_ptr():
            v.copy_(g)
        p.grad = v

    def _make_hook(self, i):
        def hook(param):
            self._settle(i)
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._launched[b]:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self._launched[b] = True
        buf = self.flat_grad[lo:hi]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self._comm_stream is not None:
            # the collective is ordered after the kernels that produced the bucket, on its own stream
            self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._comm_stream):
                self._works.append(dist.all_reduce(buf, op=op, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(buf, op=op, group=self.group, async_op=True))

    def launch_order(self):
        """bucket launch log of the current step: [(first element, last element + 1)] in the order they were sent"""
        return [self.buckets[b][:2] for b in range(len(self.buckets)) if self._launched[b]]

    def allreduce_grads(self, grads=None, async_op=False):
        """Mean of the gradients over replicas == the single-process gradient of the mean loss over the concatenated batch
        (PCCLoss is a mean over (b, c) with equal per-rank batch).  Buckets already sent from the backward hooks are only
        waited for; the rest (no hooks: `overlap=False`, a HIP-graph replay, parameters without gradient) is settled and
        sent now.  `grads`: the tensors backward wrote, when ``.grad`` no longer names them (graph replay).  With one
        replica nothing is copied or sent."""
        if self.world == 1:
            return None
        for b, (lo, hi, members) in enumerate(self.buckets):
            if self._launched[b]:
                continue
            for i in members:
                self._settle(i, None if grads is None else grads[i])
            self._launch(b)
        works, self._works = self._works, []
        if async_op:
            return works
        for w in works:
            w.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)
        if not self._avg and self.real_world > 1:
            self.flat_grad.mul_(1.0 / self.real_world)
        # ready for the next step even if the caller does not call zero_grad() (gradient accumulation into the views)
        self._pending = [len(b[2]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        return None

    # ---- captured steps (HIP graph): no Python per parameter in the hot loop ---------------------------------------------
    def set_hooks_enabled(self, on):
        """the per-bucket hooks launch collectives from inside backward: take them off while a step is captured into a graph
        (the collectives then run after the replay, allreduce_flat) and put them back for eager steps.  They are REMOVED, not
        just muted: a parameter with a post-accumulate hook is excluded from the batched end-of-backward weight-gradient
        reduction (ops._deferrable), which is legal again once nothing is sent before backward ends."""
        on = bool(on) and self.overlap
        if on and not self._hooks:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        elif not on:
            for h in self._hooks:
                h.remove()
            self._hooks = []

    def finish_capture(self):
        """Call INSIDE the capture, after backward: gradients that did not arrive in their flat-buffer view (ops without a
        destination, summed gradients) are copied there by kernels that become part of the graph; afterwards every ``.grad`` IS its
        view, so an optimizer reads the averaged values and a replay needs no per-parameter work."""
        for i in range(len(self.params)):
            self._settle(i)
        self._flat_ready = True

    def allreduce_flat(self):
        """After a graph replay whose kernels filled ``flat_grad`` (finish_capture) -- or INSIDE the capture, where the collective
        becomes a node of the graph: ONE all-reduce of the whole flat buffer, launched from the compute stream.  Nothing is left to
        overlap with, so buckets and the communication stream only add cross-stream hops: RCCL runs on the process group's own
        stream either way (compute -> RCCL -> compute, two event hops of ~10-15 us each on this stack); going through the
        communication stream as the overlapped buckets do made it four (measured on one rank at the HNOSeg-XS step: 2.745 ms against
        2.638 ms for the step without the collective; captured into the graph: 2.638 ms).  No Python per parameter."""
        if self.world == 1:
            return
        assert self._flat_ready, 'allreduce_flat() needs finish_capture() in the captured step'
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        dist.all_reduce(self.flat_grad, op=op, group=self.group)      # synchronous form: the compute stream waits for the collective
        if not self._avg and self.real_world > 1:
            self.flat_grad.mul_(1.0 / self.real_world)

    def all_ranks_ok(self, ok):
        """True iff `ok` holds on EVERY rank (one MIN all-reduce of a flag).  Decisions that change the sequence of collectives a
        rank issues -- e.g. whether a step was captured into a graph (one flat all-reduce) or runs eagerly (one all-reduce per
        bucket) -- must be taken by all ranks together, or the job hangs / reduces mismatched buffers."""
        if self.real_world == 1 or not dist.is_initialized():
            return bool(ok)
        on_gpu = self.device.type == 'cuda' and dist.get_backend(self.group) == 'nccl'
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if on_gpu else 'cpu')
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.world > 1 and self.device.type == 'cuda':
            from . import ops
            ops.set_grad_destinations(None)
            ops.set_defer_reduce(self._prev_defer)

    def __call__(self, *a, **k):
        return self.module(*a, **k)
