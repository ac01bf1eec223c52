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
        self._amp_scale = None                # torch.amp.GradScaler's scale tensor while it drives step() (see `grad_scale` below)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_mul = float(grad_scale)     # a plain multiplier on every gradient, e.g. 1 / world after a SUM all-reduce
        self._tables = {}                     # pointer signature -> (device table, rows); eager-built entries may be evicted
        self._captured_tables = {}            # the same for tables built inside a graph capture: a graph reads them at every replay,
                                              # so they live as long as the optimizer (evicting one was a use-after-free)
        self._dev = None                      # device-stepped mode: (state tensor, scheduler or None)

    # ---- torch.amp.GradScaler (round 6) ---------------------------------------------------------------------------------------
    # An optimizer with `_step_supports_amp_scaling` gets two device tensors from GradScaler.step() -- `optimizer.grad_scale` (the
    # gradients are still multiplied by it) and `optimizer.found_inf` -- instead of an unscale pass and a HOST read of the inf check
    # (torch/amp/grad_scaler.py, the contract of torch's fused optimizers).  The device-stepped kernel divides, writes the gradients
    # back unscaled and skips the update when an inf was found, so `scaler.step(optimizer); scaler.update()` of an autocast run
    # (reference experiments/train_test.py:166-168) has no host synchronisation left and fits into the captured training step.
    # Only in device-stepped mode: the host-stepped form would have to know about a skipped update to keep its step counters right.
    @property
    def _step_supports_amp_scaling(self):
        return self._dev is not None

    @property
    def grad_scale(self):
        if self._amp_scale is None:           # (GradScaler asks getattr(optimizer, 'grad_scale', 1) for a scale the user set: none)
            raise AttributeError('grad_scale')
        return self._amp_scale

    @grad_scale.setter
    def grad_scale(self, value):
        if value is None or torch.is_tensor(value):
            self._amp_scale = value
        else:                                  # the meaning this attribute had before round 6: a plain multiplier
            self.grad_mul = float(value)

    @grad_scale.deleter
    def grad_scale(self):
        self._amp_scale = None

    # ---- device-stepped mode: the whole update is capturable into a HIP graph ------------------------------------------------
    def device_stepped(self, scheduler=None):
        """Move the step counter, the learning rate and (optionally) a per-step ``CosineAnnealingWarmRestarts`` schedule (the
        reference's: experiments/run.py:92-103, stepped per batch by train_test.py:173-174) into a 9-double device state.  From now on
        ``step()`` launches hno_adamax_multi_dev: no host value changes from step to step, so the update can sit inside the captured
        training step (``CapturedStep``) and a rank's whole step is one graph replay.  Callers then do NOT call ``scheduler.step()``
        (the tick kernel advances the schedule); ``sync_from_device()`` -- called by ``state_dict()`` -- writes the counters, the
        learning rate and the scheduler's T_cur / T_i / last_epoch back to the host objects, so checkpoints keep torch's layout.
        -> True if the mode was entered (one parameter group, one common step count, supported scheduler)."""
        if self._dev is not None:
            return True
        if len(self.param_groups) != 1:
            return False
        group = self.param_groups[0]
        params = [p for p in group['params']]
        if not params or not all(p.is_cuda for p in params):
            return False
        steps = {float(self.state[p]['step']) for p in params if len(self.state[p])}
        if len(steps) > 1:
            return False
        step = steps.pop() if steps else 0.0
        st = [step, float(group['lr']), float(group['lr']), 0.0, 0.0, 1.0, 1.0, 0.0, 0.0]
        if scheduler is not None:
            from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts
            if type(scheduler) is not CosineAnnealingWarmRestarts or scheduler.optimizer is not self or len(scheduler.base_lrs) != 1:
                return False
            st[2], st[3] = float(scheduler.base_lrs[0]), float(scheduler.eta_min)
            st[4], st[5], st[6], st[7] = float(scheduler.T_cur), float(scheduler.T_i), float(scheduler.T_mult), 1.0
        # [8]: lr / (1 - beta1^(step + 1)) of the first update (the kernel's last workgroup writes the following ones), lr rounded to fp32 as
        # the eager entry point receives it; [9]: the kernel's ticket counter
        import numpy as np
        st[8] = float(np.float32(st[1])) / (1.0 - float(np.float32(group['betas'][0])) ** (step + 1.0))     # (beta1 reaches the kernels as a float)
        st.append(0.0)
        # [10]: the scheduler's step() count (its last_epoch; normally the optimizer's step count -- both are stepped once per batch --,
        # behind it only by the updates a GradScaler skipped)
        st.append(float(scheduler.last_epoch) if scheduler is not None else step)
        assert _lib.lib().hno_adamax_state_doubles() == len(st)
        self._dev = (torch.tensor(st, dtype=torch.float64, device=params[0].device), scheduler)
        # scheduler.last_epoch counts its step() calls; it normally equals the optimizer's step count (both stepped once per batch)
        self._sched_offset = int(scheduler.last_epoch) - int(step) if scheduler is not None else 0
        if not getattr(self, '_table_pool', None):
            self._reserve_tables()
        return True

    @property
    def is_device_stepped(self):
        return self._dev is not None

    def sync_from_device(self):
        """device state -> host objects (one small device-to-host copy): per-parameter ``step`` counters, the group's ``lr`` and the
        scheduler's position, exactly what the eager optimizer + scheduler.step() sequence would hold after the same number of steps."""
        if self._dev is None:
            return
        state, sched = self._dev
        v = state.cpu().tolist()
        group = self.param_groups[0]
        for p in group['params']:
            if len(self.state[p]):
                self.state[p]['step'] = torch.tensor(float(v[0]), dtype=torch.float32)
        group['lr'] = v[1]
        if sched is not None:
            sched.T_cur, sched.T_i = (int(v[4]) if float(v[4]).is_integer() else v[4]), int(v[5])
            sched._last_lr = [v[1]]
            sched.last_epoch = int(v[10])      # scheduler.step() calls so far (the kernel counts them: state [10])

    def leave_device_stepped(self):
        self.sync_from_device()
        self._dev = None

    def state_dict(self):
        self.sync_from_device()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """In device-stepped mode the loaded values are copied INTO the existing device state and moment buffers: graphs captured
        before the load keep reading the addresses they were captured with."""
        was = self._dev
        old = {}
        if was is not None:
            for p in self.param_groups[0]['params']:
                if len(self.state[p]):
                    old[p] = (self.state[p].get('exp_avg'), self.state[p].get('exp_inf'))
        self._dev = None
        super().load_state_dict(state_dict)
        if was is not None:          # re-enter the mode on the loaded counters (the caller reloads the scheduler first)
            with torch.no_grad():
                for p, bufs in old.items():
                    st = self.state[p]
                    for k, buf in zip(('exp_avg', 'exp_inf'), bufs):
                        if buf is not None and k in st and torch.is_tensor(st[k]) and st[k].shape == buf.shape:
                            buf.copy_(st[k])
                            st[k] = buf
            if self.device_stepped(was[1]) and self._dev[0] is not was[0]:
                was[0].copy_(self._dev[0])
                self._dev = (was[0], self._dev[1])

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
        hit = self._captured_tables.get(key) or self._tables.get(key)
        if hit is not None:
            return hit
        rows = []
        for p, g, m, u in tensors:
            n = p.numel()
            for off in range(0, n, _CHUNK):
                rows.append([p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off, u.data_ptr() + 4 * off,
                             min(_CHUNK, n - off)])
        assert _lib.lib().hno_adamax_chunk_rows() == 5
        dev = tensors[0][0].device
        if torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture nothing may be allocated or pinned: take a (pinned host, device) buffer pair set aside by
            # device_stepped(); the copy becomes a node of the graph that re-reads the pinned rows at every replay, so a pair is
            # written once and never reused
            pool = getattr(self, '_table_pool', [])
            if not pool or pool[-1][0].shape[0] < len(rows):
                raise _lib.HnoError('optim.Adamax: no pre-allocated table left for a captured step (run this batch shape eagerly)')
            host, table = pool.pop()
            host[:len(rows)] = torch.tensor(rows, dtype=torch.int64)
            # the rows reach the device ONCE, after the capture (finish_capture): a copy node inside the graph would move the same
            # 8 KB again at every replay (4.7 us per step in the kernel trace of round 4)
            self._unfilled = getattr(self, '_unfilled', []) + [(host, table)]
            self._captured_tables[key] = (table, len(rows))
            return self._captured_tables[key]
        table = torch.tensor(rows, dtype=torch.int64).to(dev)
        if len(self._tables) > 8:
            self._tables.clear()
        self._tables[key] = (table, len(rows))
        return self._tables[key]

    def finish_capture(self):
        """after a graph capture that contained step(): copy the chunk tables built during the capture to the device (synchronous; the
        captured launch reads them at every replay).  MUST run before the first replay -- CapturedStep and bench.py call it."""
        for host, table in getattr(self, '_unfilled', []):
            table.copy_(host)
        self._unfilled = []
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def _reserve_tables(self, count=4):
        """(pinned host, device) buffer pairs for chunk tables built inside graph captures (gradient addresses of a captured step are
        only known while it is being captured)"""
        params = self.param_groups[0]['params']
        nrows = sum(-(-p.numel() // _CHUNK) for p in params)
        dev = params[0].device
        self._table_pool = [(torch.zeros((nrows, 5), dtype=torch.int64).pin_memory(), torch.zeros((nrows, 5), dtype=torch.int64, device=dev))
                            for _ in range(count)]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        if getattr(self, '_unfilled', None) and not torch.cuda.is_current_stream_capturing():
            self.finish_capture()
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
                if self._dev is not None:
                    by_step.setdefault(0, []).append((p, p.grad, st['exp_avg'], st['exp_inf']))
                    continue
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
                amp_scale, found_inf = self._amp_scale, getattr(self, 'found_inf', None)
                if self._dev is not None:
                    if amp_scale is not None or found_inf is not None:      # driven by torch.amp.GradScaler
                        for t_ in (amp_scale, found_inf):
                            if t_ is not None and not (t_.is_cuda and t_.dtype == torch.float32 and t_.numel() == 1):
                                raise _lib.HnoError('optim.Adamax: grad_scale / found_inf must be fp32 device scalars')
                        check(L.hno_adamax_multi_dev_amp(table.data_ptr(), nrows, self._dev[0].data_ptr(), float(beta1), float(beta2),
                                                         float(group['eps']), float(group['weight_decay']), self.grad_mul,
                                                         None if amp_scale is None else amp_scale.data_ptr(),
                                                         None if found_inf is None else found_inf.data_ptr(), stream_ptr()),
                              'hno_adamax_multi_dev_amp')
                        continue
                    check(L.hno_adamax_multi_dev(table.data_ptr(), nrows, self._dev[0].data_ptr(), float(beta1), float(beta2),
                                                 float(group['eps']), float(group['weight_decay']), self.grad_mul, stream_ptr()),
                          'hno_adamax_multi_dev')
                    continue
                if amp_scale is not None or found_inf is not None:
                    raise _lib.HnoError('optim.Adamax: GradScaler tensors reach step() only in device-stepped mode')
                check(L.hno_adamax_multi(table.data_ptr(), nrows, float(group['lr']), float(beta1), float(beta2),
                                         float(group['eps']), float(group['weight_decay']), t, self.grad_mul,
                                         stream_ptr()), 'hno_adamax_multi')
        return loss
