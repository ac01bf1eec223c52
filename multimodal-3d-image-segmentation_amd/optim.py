"""Fused optimizer for the training loop.

``Adamax`` is a drop-in for ``torch.optim.Adamax`` as the reference constructs it
(experiments/run.py:89-91, ``[optimizer] optimizer_name = 'Adamax', lr = 5e-3``): same constructor
arguments, same ``param_groups`` (so ``torch.optim.lr_scheduler.CosineAnnealingWarmRestarts``, stepped
per batch by train_test.py:173-174, drives it unchanged) and the same ``state_dict`` layout
(per-parameter ``step`` / ``exp_avg`` / ``exp_inf``), so checkpoints written with either load into
the other.  ``step()`` is ONE HIP launch over all parameters (hno_adamax_multi) instead of torch's
~10 multi-tensor launches; there is no CPU path.
"""
import torch

from . import _lib
from ._lib import check, stream_ptr

_CHUNK = 4096   # elements per workgroup (256 threads x 16)


class Adamax(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        if lr < 0.0:
            raise ValueError(f'Invalid learning rate: {lr}')
        if eps < 0.0:
            raise ValueError(f'Invalid epsilon value: {eps}')
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f'Invalid beta parameter at index 0: {betas[0]}')
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f'Invalid beta parameter at index 1: {betas[1]}')
        if weight_decay < 0.0:
            raise ValueError(f'Invalid weight_decay value: {weight_decay}')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)   # e.g. 1 / world after a SUM all-reduce
        self._tables = {}                     # pointer signature -> (device table, rows)

    def _init_state(self, group):
        """exp_avg / exp_inf of a group live in two flat buffers (views per parameter)."""
        new = [p for p in group['params'] if p.grad is not None and len(self.state[p]) == 0]
        if not new:
            return
        n = sum(p.numel() for p in new)
        flat = torch.zeros(2 * n, device=new[0].device, dtype=torch.float32)
        off = 0
        for p in new:
            st = self.state[p]
            st['step'] = torch.tensor(0.0, dtype=torch.float32)
            st['exp_avg'] = flat[off:off + p.numel()].view_as(p)
            st['exp_inf'] = flat[n + off:n + off + p.numel()].view_as(p)
            off += p.numel()

    def _table(self, tensors):
        key = tuple(t.data_ptr() for quad in tensors for t in quad)
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        rows = []
        for p, g, m, u in tensors:
            n = p.numel()
            for off in range(0, n, _CHUNK):
                rows.append([p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off, u.data_ptr() + 4 * off,
                             min(_CHUNK, n - off)])
        assert _lib.lib().hno_adamax_chunk_rows() == 5
        table = torch.tensor(rows, dtype=torch.int64).to(tensors[0][0].device)
        if len(self._tables) > 8:
            self._tables.clear()
        self._tables[key] = (table, len(rows))
        return self._tables[key]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            self._init_state(group)
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('Adamax does not support sparse gradients')
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and p.grad.dtype == torch.float32):
                    raise _lib.HnoError('the fused Adamax needs contiguous fp32 parameters and gradients on the GPU '
                                        '(there is no CPU fallback)')
                st = self.state[p]
                for k in ('exp_avg', 'exp_inf'):   # e.g. after load_state_dict from a CPU checkpoint
                    if st[k].device != p.device or st[k].dtype != torch.float32 or not st[k].is_contiguous():
                        st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
                if torch.is_tensor(st['step']) and st['step'].device.type != 'cpu':
                    # after load_checkpoint(map_location=device) the counters sit on the GPU: `+= 1` would launch a kernel and
                    # int() would synchronise, per parameter and step.  Keep them on the host (torch's non-capturable layout).
                    st['step'] = st['step'].detach().to('cpu', torch.float32)
                st['step'] += 1
                by_step.setdefault(int(st['step']), []).append((p, p.grad, st['exp_avg'], st['exp_inf']))
            beta1, beta2 = group['betas']
            for t, tensors in by_step.items():
                table, nrows = self._table(tensors)
                check(L.hno_adamax_multi(table.data_ptr(), nrows, float(group['lr']), float(beta1), float(beta2),
                                         float(group['eps']), float(group['weight_decay']), t, self.grad_scale,
                                         stream_ptr()), 'hno_adamax_multi')
        return loss
