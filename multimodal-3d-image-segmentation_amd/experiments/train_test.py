"""`training()` with the reference's signature and semantics (experiments/train_test.py:31-286).

Kept: argument list, checkpoint/resume protocol and file names (`model/model.pt`,
`model/checkpoint.pt`, `stdout.txt` with the `train_loss:` / `valid_loss:` line format), the order
forward -> zero_grad -> backward -> step, the per-BATCH scheduler step, the mean of per-batch losses,
the checkpoint cadence and the best-model selection rule.
Changed on purpose: labels go to the loss kernels as uint8 class maps (no one-hot tensor), and the
per-batch `loss.item()` host synchronisation (reference :162) is deferred to the end of the epoch.
Out of scope: torchview graph rendering, matplotlib plots.
"""
import contextlib
import os
import re
import sys
import time
from os.path import join

import numpy as np
import torch

from .. import ops
from ..ops import backward_from
from .utils import labels_to_u8, save_model_summary


step_stats = {'replayed': 0, 'eager': 0}     # training steps by launch form (read by the tests)
last_run = {}                                # what the last training() call did: device-stepped optimizer, measured schedules (tests, logs)


class SampleSplit:
    """The two halves of a batch as two concurrent forward + loss + backward passes on two streams of the capturing graph (round 4b).

    The samples of a batch are independent (no batch statistics anywhere in these models) and the losses are means over (sample, class),
    so loss = (loss_A + loss_B) / 2 and every gradient is the sum of the halves' gradients.  Run one after the other the halves gain
    nothing; run on two streams INSIDE one captured graph the latency-bound kernels of one half (the fused spectral middles, the small
    reductions: ~0.4 ms of the 2.4 ms HNOSeg-XS step keep three quarters of the chip idle) execute under the bandwidth-bound kernels of
    the other: 2.41 -> 2.32 ms measured with two independent models (tools/dbg/two_stream.py, DESIGN lesson 57).
    The second half runs through a TWIN of the model -- a deep copy whose parameters and buffers alias the model's storage, so it always
    computes with the current weights but collects its gradients in its own ``.grad`` tensors; ONE libhno launch (ops.sum_pairs) joins
    the gradient sets and averages the two losses.
    Only used inside captured steps (CapturedStep / bench.py); eager steps run the whole batch as before.  Whether a captured step takes
    this schedule is MEASURED when the step is captured (``choose_schedule``: both forms are captured, replayed three times each, the
    faster one is kept) for model classes that name themselves candidates (``hno_sample_split = 'measure'``: HNOSegXS);
    HNO_SPLIT_STREAMS=1 / 0 forces it on / off for every model.  Build it OUTSIDE the capture (its deep copy and constants are not part
    of the step)."""

    _streams = {}

    def __init__(self, model):
        import copy
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('build a SampleSplit before the capture: its copies would become nodes of the graph')
        self.model = model
        # tensor hooks stay with the model (a data-parallel replica's post-accumulate hooks would be deep-copied together with the
        # replica they point to, and the twin's backward would then launch that copy's bucket all-reduces)
        stash = []
        for p in model.parameters():
            stash.append((p, getattr(p, '_post_accumulate_grad_hooks', None), getattr(p, '_backward_hooks', None)))
            if stash[-1][1] is not None:
                p._post_accumulate_grad_hooks = None
            if stash[-1][2] is not None:
                p._backward_hooks = None
        try:
            self.twin = copy.deepcopy(model)
        finally:
            for p, post, back in stash:
                if post is not None:
                    p._post_accumulate_grad_hooks = post
                if back is not None:
                    p._backward_hooks = back
        for p, q in zip(model.parameters(), self.twin.parameters()):
            q.data = p.data
            q.grad = None
        for a, b in zip(model.buffers(), self.twin.buffers()):
            b.data = a.data
        self.twin.train(model.training)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.tparams = [q for q in self.twin.parameters() if q.requires_grad]
        dev = self.params[0].device
        self.half = torch.full((), 0.5, device=dev, dtype=torch.float32)
        self.outputs = None
        # one pair of streams per device for every split of the process: the gradient-accumulation nodes autograd keeps per parameter
        # remember the stream they were created on, and a node that outlives one capture must not name a stream the next capture does
        # not know (a stream outside the capture being waited on inside it ends the capture with a crash)
        key = (dev.type, dev.index)
        if key not in SampleSplit._streams:
            SampleSplit._streams[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        self.streams = SampleSplit._streams[key]
        torch.cuda.synchronize(dev)

    @staticmethod
    def candidate(model, loss_fn, x):
        """-> False, 'force' or 'measure'.  The schedule is possible for an even batch of independent samples under a loss that is a mean
        over (sample, class); it is WANTED when HNO_SPLIT_STREAMS=1 ('force'), or when the model class names itself a candidate
        (``hno_sample_split``: True = 'force', 'measure' = let choose_schedule() time both forms of the step at capture time).  Round 4
        carried a list of shapes measured by hand (HNOSeg-XS gains 3-6 % at 2 x 4 x 128^3 and loses 2-10 % at 80^3 ... 160^3; FNOSeg loses
        1 %): a constant in model code, replaced by the measurement."""
        env = os.environ.get('HNO_SPLIT_STREAMS', '')
        if env == '0':
            return False
        spec = 'force' if env == '1' else getattr(model, 'hno_sample_split', False)
        if spec is True:
            spec = 'force'
        if spec not in ('force', 'measure'):
            return False
        if not (torch.is_tensor(x) and x.is_cuda) or x.shape[0] < 2 or x.shape[0] % 2:
            return False
        if getattr(loss_fn, 'hno_loss_spec', None) is None:      # a mean over (sample, class): halves average exactly
            return False
        norm = (torch.nn.modules.batchnorm._BatchNorm,)
        if any(isinstance(m, norm) for m in model.modules()) or not all(p.is_cuda for p in model.parameters()):
            return False
        return spec

    @staticmethod
    def usable(model, loss_fn, x):
        return bool(SampleSplit.candidate(model, loss_fn, x))

    def aliased(self):
        """the twin still shares the model's storage (a model moved or re-materialised after the split was built does not)"""
        return all(p.data_ptr() == q.data_ptr() for p, q in zip(self.model.parameters(), self.twin.parameters()))

    def fwd_bwd(self, x, lab_u8, loss_fn, zero_grad=None, autocast=None, keep_outputs=False):
        """enqueue both halves (call with the capturing stream current) -> the batch loss; model parameters' .grad = full gradients.
        Only inside a stream capture (there every tensor of the step lives in the graph's private pool until the capture ends): eager, the
        two passes are launch-bound and gain nothing, and the caching allocator's per-stream reuse would need record_stream bookkeeping.
        keep_outputs: leave the halves' (detached) model outputs in ``self.outputs`` (tests read them after a replay)."""
        if not torch.cuda.is_current_stream_capturing():
            raise RuntimeError('SampleSplit.fwd_bwd is for captured steps only (CapturedStep / bench.py)')
        cur = torch.cuda.current_stream()
        h = x.shape[0] // 2
        losses, outs = [], []
        for m, s, sl in ((self.model, self.streams[0], slice(0, h)), (self.twin, self.streams[1], slice(h, x.shape[0]))):
            s.wait_stream(cur)
            with torch.cuda.stream(s), (autocast() if autocast is not None else contextlib.nullcontext()):
                xb, lb = x[sl], lab_u8[sl]
                with ops.expected_loss(lb, loss_fn):
                    y = m(xb)
                losses.append(loss_fn(y, lb))
                if keep_outputs:
                    outs.append(y.detach())
                del y
        self.outputs = outs if keep_outputs else None
        if zero_grad is not None:
            zero_grad()
        else:
            for p in self.params:
                p.grad = None
        for q in self.tparams:
            q.grad = None
        for l, s in zip(losses, self.streams):
            s.wait_stream(cur)          # zero_grad may have launched on the capturing stream (a replica clears its flat gradient buffer)
            with torch.cuda.stream(s):
                torch.autograd.backward(l, grad_tensors=self.half)
        cur.wait_stream(self.streams[0])
        cur.wait_stream(self.streams[1])
        pairs = [(p, q) for p, q in zip(self.params, self.tparams) if q.grad is not None]
        for p, q in pairs:
            if p.grad is None:
                p.grad = q.grad
        # (detached: a loss that keeps its autograd graph alive keeps the halves' accumulation nodes alive across captures)
        l0, l1 = losses[0].detach().float().contiguous(), losses[1].detach().float().contiguous()
        del losses
        loss = torch.empty_like(l0)
        # the join: every gradient of the model += its twin's, loss = (l0 + l1) / 2 -- ONE libhno launch (round 4: torch._foreach_add_ +
        # torch.lerp, the only two ATen kernels of the step)
        ops.sum_pairs([(p.grad, p.grad, q.grad, 1.0) for p, q in pairs if p.grad is not q.grad] + [(loss, l0, l1, 0.5)])
        return loss


def vote_on_schedule(measure, agree=None):
    """``measure()`` -> [ms_one_pass, ms_two_streams]; -> (use_split, times).  The agreement is a COLLECTIVE (a MIN all-reduce over the
    ranks): every rank must reach it whatever happened locally.  A rank whose capture or timing raised (graph-pool OOM, a transient HIP
    error) votes False and re-raises AFTER the vote -- were it to skip the vote, the other ranks' all-reduce would pair with this rank's
    next, different collective and every later one would be off by one: the hang the vote exists to prevent."""
    times, failure = [float('inf'), float('inf')], None
    try:
        times = list(measure())
    except Exception as exc:        # noqa: BLE001 -- vote first, raise afterwards
        failure = exc
    use = failure is None and times[1] < times[0]
    if agree is not None:
        use = bool(agree(use))
    if failure is not None:
        raise failure
    return use, times


def choose_schedule(run_one, run_split, replays=5, rounds=3, agree=None, what=''):
    """Measure, do not guess: capture the step's forward + loss + backward in both forms -- one pass over the batch (``run_one()``) and
    the two half-batches on two streams (``run_split()``) -- into two temporary graphs, warm both up, time ``rounds`` alternating bursts
    of ``replays`` replays each (HIP events; the minimum per form: clocks and caches settle during the first bursts -- a single cold burst
    of three replays read 2.49 ms for both forms of a step whose steady state is 2.35 / 2.27 ms), free the graphs, and return
    (use_split, ms_one, ms_split).  Nothing but forward / loss / backward may be in the closures (no optimizer, no collective: the
    replays must not change any state but gradients, which the caller resets).
    ``agree``: a callable flag -> flag that makes ranks agree (FlatGradReplica.all_ranks_ok): a rank-local timing decision would give
    the ranks different graphs."""
    def measure():
        times = [float('inf'), float('inf')]
        graphs = []
        for fn in (run_one, run_split):
            cur = torch.cuda.current_stream()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                    out = fn()
                del out
            cur.wait_stream(side)
            torch.cuda.synchronize()
            graphs.append(graph)
        for graph in graphs:
            for _ in range(3):
                graph.replay()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for i, graph in enumerate(graphs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(replays):
                    graph.replay()
                e1.record()
                torch.cuda.synchronize()
                times[i] = min(times[i], e0.elapsed_time(e1) / replays)
        return times

    use, times = vote_on_schedule(measure, agree)
    if os.environ.get('HNO_TRAIN_GRAPH_QUIET', '0') == '0':
        print(f'[hno] captured step{what}: one pass over the batch {times[0]:.3f} ms, two half-batches on two streams {times[1]:.3f} ms '
              f'-> {"two streams" if use else "one pass"}', file=sys.stderr, flush=True)
    return use, times[0], times[1]


class CapturedStep:
    """forward + loss + backward of one batch shape as a HIP graph -- the step bench.py replays, for `training()`.

    An eager step costs the host ~125 kernel launches; replayed from a graph the same kernels run back to back (HNOSeg-XS, 2 x 4 x
    128^3: 3.4 ms eager, 2.65 ms replayed).  A shape is captured on its SECOND occurrence (the first runs eagerly: it creates the
    twiddle tables and kernel attributes, which cannot be captured); at most `max_shapes` shapes are kept (the ragged last batch of
    an epoch then simply runs eagerly).  After a replay every parameter's ``.grad`` is the buffer the graph wrote, whatever eager
    steps did in between.  Autocast runs (round 6): forward + loss + scaled backward are replayed; GradScaler's step / update join the
    replay when the optimizer takes the scale and the inf flag as device tensors (optim.Adamax, device-stepped), otherwise they stay eager
    behind it (their inf check synchronises).  Not used on CPU tensors."""

    def __init__(self, model, loss_fn, num_labels, label_mapping=None, data_parallel=None, max_shapes=2, optimizer=None, bucketed=None,
                 autocast=None, scaler=None):
        """optimizer: a device-stepped optim.Adamax (``optimizer.device_stepped(scheduler)``): its update (and the scheduler's step)
        is captured behind backward -- and behind the gradient all-reduce, which is then captured too -- so a step of a rank is ONE
        graph replay; ``steps_optimizer`` tells the caller not to step again.
        With replicas over SEVERAL ranks the collective is captured only on request (HNO_DP_CAPTURE_ALLREDUCE=1): a collective inside
        a graph has never run on more than one GPU from this pool, and a mismatch there hangs instead of raising;
        the default is the measured form -- replay, then one eager flat all-reduce, then the eager Adamax launch."""
        self.model, self.loss_fn, self.num_labels, self.label_mapping = model, loss_fn, num_labels, label_mapping
        self.dp, self.max_shapes = data_parallel, max_shapes
        # autocast (round 6): a callable returning the autocast context (training(use_autocast=True)) -- forward and loss are captured
        # inside it, and the backward starts from scaler.scale(loss) (the scale is a device tensor the scaler updates in place, so every
        # replay reads the current one).  scaler.step() / update() stay eager behind the replay: their inf check is a host
        # synchronisation, which is why round 3 left autocast runs eager altogether (FNOSeg cfg3: 7.6 ms eager, 5.2 ms replayed).
        self.autocast, self.scaler = autocast, scaler
        if autocast is not None and not (scaler is not None and getattr(optimizer, '_step_supports_amp_scaling', False)):
            optimizer = None                 # the update belongs to an eager scaler.step()
        # (round 6b: a device-stepped optim.Adamax takes GradScaler's scale and inf flag as device tensors, so scaler.step(optimizer) and
        # scaler.update() have no host synchronisation and are captured behind the scaled backward: an autocast step is ONE replay too)
        self.optimizer = optimizer if (optimizer is not None and getattr(optimizer, 'is_device_stepped', False)) else None
        env = os.environ.get('HNO_DP_CAPTURE_ALLREDUCE', '')
        multi_rank = data_parallel is not None and getattr(data_parallel, 'real_world', 1) > 1
        # the optimizer can only follow the all-reduce: with replicas the collective has to be part of the graph as well
        self.capture_allreduce = data_parallel is not None and (env == '1' or (env != '0' and not multi_rank and self.optimizer is not None))
        if data_parallel is not None and not self.capture_allreduce:
            self.optimizer = None
        self.steps_optimizer = self.optimizer is not None
        # bucketed (round 6): with the collective inside the graph AND a gradient worth overlapping (V-Net-DS: 90 MB), the per-bucket
        # all-reduces are launched by the replica's post-accumulate hooks DURING the captured backward, on its communication stream --
        # they become nodes of the graph on a side branch and overlap the rest of backward in every replay, as the eager path does.
        # Default: when the flat gradient is at least HNO_DP_BUCKETED_BYTES (4 MB); HNOSeg-XS's 113 KB stay one flat all-reduce.
        big = data_parallel is not None and data_parallel.flat_grad.numel() * 4 >= int(os.environ.get('HNO_DP_BUCKETED_BYTES', str(4 << 20)))
        self.bucketed = bool(self.capture_allreduce and getattr(data_parallel, 'overlap', False) and (big if bucketed is None else bucketed))
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.entries, self.seen, self.failed = {}, {}, set()
        self.copy_stream, self.staged = None, None
        self.split = None
        self.keep_outputs = False
        self.schedule = {}          # batch-shape key -> ('one pass' | 'two streams', ms_one, ms_split)

    def _fwd_bwd(self, xs, ys, use_split):
        """the step's forward + loss + backward, enqueued on the current (capturing) stream -> loss"""
        from .. import ops
        lab = labels_to_u8(ys, self.num_labels, self.label_mapping)
        if use_split:
            return self.split.fwd_bwd(xs, lab, self.loss_fn, zero_grad=self.dp.zero_grad if self.dp is not None else None,
                                      keep_outputs=self.keep_outputs)
        with (self.autocast() if self.autocast is not None else contextlib.nullcontext()):
            with ops.expected_loss(lab, self.loss_fn):       # the head takes the loss sums in its own pass
                y_pred = self.model(xs)
            loss = self.loss_fn(y_pred, lab)
        if self.keep_outputs:
            self.outputs = [y_pred.detach()]
        if self.dp is not None:
            self.dp.zero_grad()
        else:
            for p in self.params:
                p.grad = None
        if self.scaler is not None:
            self.scaler.scale(loss).backward()      # reference train_test.py:166
        else:
            ops.backward_from(loss)
        return loss.detach()

    def _capture(self, x, y):
        from .. import ops
        xs, ys = x.clone(), y.clone()
        # (bucketed: the twin's gradients bypass the hooks; autocast: one scaled loss starts the backward)
        mode = False if (self.bucketed or self.autocast is not None) else SampleSplit.candidate(self.model, self.loss_fn, xs)
        if mode:
            # the twin, its constants and the half-batch shapes' tables / kernel attributes exist BEFORE the capture (none of it is
            # capturable; round 4 built the twin inside the capture: its copies were replayed with every step -- ADVICE round 4)
            try:
                if self.split is None or not self.split.aliased():
                    self.split = SampleSplit(self.model)
                with torch.no_grad():
                    self.model(xs[:xs.shape[0] // 2])
            except Exception:                # a model that cannot be deep-copied: one pass over the batch
                self.split, mode = None, False
        if self.dp is not None and SampleSplit.candidate(self.model, self.loss_fn, xs):
            # whether the twin could be built is rank-local, whether choose_schedule()'s collective runs must not be: every rank of a
            # candidate shape votes, and one rank without a twin takes the split (and the measurement) away from all of them
            if not self.dp.all_ranks_ok(bool(mode)):
                self.split, mode = None, False
        cur = torch.cuda.current_stream()
        torch.cuda.synchronize()
        if self.dp is not None:
            self.dp.set_hooks_enabled(self.bucketed)
        # inside a captured step nothing reads a gradient before backward ends -- unless its buckets leave during backward
        prev = ops.set_defer_reduce(not self.bucketed)
        key = (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype)
        try:
            use_split = mode == 'force'
            if mode == 'measure':
                use_split, ms_one, ms_split = choose_schedule(lambda: self._fwd_bwd(xs, ys, False), lambda: self._fwd_bwd(xs, ys, True),
                                                              agree=self.dp.all_ranks_ok if self.dp is not None else None,
                                                              what=f' for batches of shape {tuple(x.shape)}')
                self.schedule[key] = ('two streams' if use_split else 'one pass', ms_one, ms_split)
                for p in self.params + self.split.tparams:      # the temporary graphs' gradient buffers are gone
                    p.grad = None
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                # thread_local: a collective library's watchdog thread may poll events while we capture
                with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                    loss = self._fwd_bwd(xs, ys, use_split)
                    if self.dp is not None and self.bucketed:
                        # the hooks sent the buckets on the communication stream (a side branch of the graph); what is left is
                        # sent now, the capturing stream joins the branch
                        self.dp.allreduce_grads()
                        self.dp._flat_ready = True
                    elif self.dp is not None:
                        self.dp.finish_capture()
                        if self.capture_allreduce:
                            self.dp.allreduce_flat()
                    if self.optimizer is not None:
                        if self.scaler is not None:
                            self.scaler.step(self.optimizer)      # reference train_test.py:167-168
                            self.scaler.update()
                        else:
                            self.optimizer.step()
            cur.wait_stream(side)
            torch.cuda.synchronize()
            if self.optimizer is not None:
                self.optimizer.finish_capture()      # chunk tables built during the capture -> device, once
        except Exception as exc:        # capture not possible for this model / shape: stay eager
            if os.environ.get('HNO_TRAIN_GRAPH_DEBUG'):
                import traceback
                traceback.print_exc()
            ops.set_defer_reduce(prev)
            self.split = None           # (a twin whose gradients point into a dropped graph's pool must not be reused)
            if self.dp is not None:
                self.dp.set_hooks_enabled(True)
            return None
        ops.set_defer_reduce(prev)
        if self.dp is not None and self.bucketed:
            self.dp.set_hooks_enabled(False)
        # (data-parallel: the hooks stay off while graphs exist; eager fall-back steps send their buckets from allreduce_grads())
        # staging buffers of prefetch(): the NEXT batch crosses PCIe on the copy stream while this one is being worked on
        done = torch.cuda.Event()
        done.record(cur)
        return graph, xs, ys, loss, [p.grad for p in self.params], torch.empty_like(xs), torch.empty_like(ys), [done]

    def prefetch(self, x, y):
        """Start moving the NEXT batch (host tensors) to the device on a copy stream; call it right after step() has launched the
        current batch.  The graph's input buffers are still being read (conv_in's weight gradient reads the image at the very end of
        backward), so the copy lands in per-shape staging buffers and step() moves it over with a device-to-device copy.  PCIe-inclusive
        rate of the HNOSeg-XS step with both copies serialised on the compute stream: 4.5 ms per step; overlapped: see DESIGN section 5."""
        self.staged = None
        if x is None or x.is_cuda:
            return
        ent = self.entries.get((tuple(x.shape), x.dtype, tuple(y.shape), y.dtype))
        if ent is None:
            return
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream()
        xst, yst, done = ent[5], ent[6], ent[7]
        self.copy_stream.wait_event(done[0])         # the last device-to-device copy out of the staging buffers has run
        with torch.cuda.stream(self.copy_stream):
            xst.copy_(x, non_blocking=True)
            yst.copy_(y, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.staged = (x, y, ev)

    def step(self, x, y):
        """-> the (static) loss tensor after one replayed forward + loss + backward, or None: run this batch eagerly.  x, y may still
        be host tensors: they are copied straight into the graph's input buffers."""
        key = (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype)
        ent = self.entries.get(key)
        if ent is None:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if n < 1 or key in self.failed or len(self.entries) >= self.max_shapes:
                return None
            dev = self.params[0].device
            ent = self._capture(x.to(dev), y.to(dev))
            if self.dp is not None and not self.dp.all_ranks_ok(ent is not None):
                # a capture that failed on ANY rank (graph-pool OOM, a transient HIP error) is dropped on EVERY rank: a rank that
                # replays sends one flat all-reduce, an eager rank one per bucket -- mixed, the collectives no longer match
                if ent is not None:
                    ent = None
                    self.split = None
                    if not self.entries:
                        self.dp.set_hooks_enabled(True)
            if ent is None:
                self.failed.add(key)
                return None
            self.entries[key] = ent
            if os.environ.get('HNO_TRAIN_GRAPH_QUIET', '0') == '0' and (self.dp is None or torch.distributed.get_rank(self.dp.group) == 0):
                print(f'training(): batches of shape {tuple(x.shape)} are replayed from a HIP graph from now on '
                      f'(module / loss hooks and host-side logic do not run in replayed steps; use_graph=False keeps every step eager)')
        graph, xs, ys, loss, grads, xst, yst, done = ent
        st = self.staged
        if st is not None and st[0] is x and st[1] is y:      # prefetched: already on the device
            cur = torch.cuda.current_stream()
            cur.wait_event(st[2])
            xs.copy_(xst, non_blocking=True)
            ys.copy_(yst, non_blocking=True)
            done[0] = torch.cuda.Event()
            done[0].record(cur)
            self.staged = None
        else:
            xs.copy_(x, non_blocking=True)
            ys.copy_(y, non_blocking=True)
        graph.replay()
        for p, g in zip(self.params, grads):
            p.grad = g
        if self.dp is not None and not self.capture_allreduce:
            self.dp.allreduce_flat()
        return loss


def training(model, input_data, output_dir, loss_fn, optimizer, scheduler=None, label_mapping=None, num_epochs=100,
             selection_epoch_portion=0.8, checkpoint_epoch=10, is_plot_model=False, is_print=True,
             plot_epoch_portion=None, use_autocast=False, device=None, data_parallel=None, use_graph=None):
    """Trains a model; see the reference docstring (train_test.py:48-71).  `data_parallel` is an optional
    parallel.FlatGradReplica for one-process-per-GPU training (not in the reference).  `use_graph` (not in the reference): replay
    forward + loss + backward of recurring batch shapes from a HIP graph (CapturedStep); None = on for this package's model families on
    CUDA (HNO_TRAIN_GRAPH=0 switches it off; autocast runs since round 6: HNO_TRAIN_GRAPH_AUTOCAST=0 keeps those eager); the same kernels either way.  What changes with replay (logged once per
    captured shape): forward / backward hooks and any host-side logic inside the model or loss_fn do not run in replayed steps; each
    captured shape (at most 2) keeps a private graph memory pool plus input / staging buffers for the rest of training, which eager
    validation cannot reuse (a model close to the memory limit may prefer use_graph=False); under data_parallel the per-bucket
    overlap hooks stay off once a shape is captured (eager fall-back steps then send their buckets after backward)."""
    import contextlib
    import torch.distributed as dist
    # use_autocast (reference :79, :154-168): forward + loss under autocast, GradScaler around backward / step, its state in
    # the checkpoint.  The matrix-core path here is bfloat16 (the reference leaves torch's per-device default dtype: float16
    # on CUDA, bfloat16 on CPU); models switch to their bf16 kernels inside the context (ops_bf16.autocast_bf16()).
    scaler = torch.amp.GradScaler(device='cuda') if use_autocast else None

    def autocast():
        return torch.autocast(device_type='cuda', dtype=torch.bfloat16) if use_autocast else contextlib.nullcontext()
    # one process per GPU (data_parallel given): rank 0 alone logs, writes the summary and saves; every rank resumes from the
    # same checkpoint; the validation loss is averaged over ranks so that all ranks select the same best epoch
    world = data_parallel.world if data_parallel is not None else 1
    rank = dist.get_rank(data_parallel.group) if world > 1 else 0
    is_main = rank == 0
    is_print = is_print and is_main
    model_dir = join(output_dir, 'model')
    model_path, chkpt_path = join(model_dir, 'model.pt'), join(model_dir, 'checkpoint.pt')
    stdout_file = join(output_dir, 'stdout.txt')
    if is_main:
        os.makedirs(model_dir, exist_ok=True)

    def barrier():
        if world > 1:
            dist.barrier(group=data_parallel.group)

    def log(*lines, echo=True):
        if not is_main:
            return
        if is_print and echo:
            for ln in lines:
                print(ln)
        with open(stdout_file, 'a') as f:
            for ln in lines:
                print(ln, file=f)

    model.to(device)
    barrier()                       # the directory (and a checkpoint written by an earlier run) is visible to every rank
    if os.path.exists(chkpt_path):
        start_epoch, min_loss, best_epoch = load_checkpoint(chkpt_path, model, optimizer, scheduler, scaler, device)
        start_epoch += 1
        if start_epoch >= num_epochs:
            raise RuntimeError(f'Checkpoint detected, but start_epoch ({start_epoch}) >= num_epochs ({num_epochs})')
        if is_print:
            print(f'Checkpoint loaded for epoch {start_epoch}')
        barrier()                   # everyone has read the checkpoint before rank 0 rewrites files
        if is_main:
            # drop what was logged after the last checkpoint (reference :92-102)
            with open(stdout_file) as f:
                lines = f.readlines()
            last = max((i for i, ln in enumerate(lines) if 'checkpoint' in ln), default=len(lines) - 1)
            with open(stdout_file, 'w') as f:
                f.writelines(lines[:last + 1])
    else:
        start_epoch, min_loss, best_epoch = 0, float('inf'), None
        log('', f'train_num_batches: {input_data.get_train_num_batches()}',
            f'valid_num_batches: {input_data.get_valid_num_batches()}', '')
        if is_main:
            # shape-only forward of a copy of the model on the meta device (reference :117-119)
            input_size = (1, model.in_channels) + tuple(input_data.get_train_image_size())
            save_model_summary(model, input_size, join(output_dir, 'model_summary.txt'))
    if world > 1 and hasattr(input_data, 'set_shard'):
        input_data.set_shard(rank, world)     # disjoint, equally sized shards of every epoch's (shared-seed) permutation

    train_flow = input_data.get_train_flow(shuffle=True)
    valid_flow = input_data.get_valid_flow()
    num_labels = model.out_channels
    if is_print:
        print('Training started')
        print(output_dir)
    start_time = time.time()

    def mean_loss(losses):
        return float(np.mean([float(v) for v in torch.stack(losses).cpu()])) if losses else float('nan')

    if use_graph is None:
        # automatic for this package's four model families, whose captured steps reproduce the eager trajectories bit for bit
        # (G8 golden, tools/dbg/graph_train_ab.py, bench.py); anything else (user modules with host-side state) only on request
        from ..nets.hnosegxs import HNOSegXS
        from ..nets.architectures import NeuralOperatorSeg, HartleyMHASeg, VNetDS
        use_graph = os.environ.get('HNO_TRAIN_GRAPH', '1') != '0' and isinstance(model, (HNOSegXS, NeuralOperatorSeg, HartleyMHASeg, VNetDS))
    captured, entered_dev_opt = None, False
    if use_graph and use_autocast and next(model.parameters()).is_cuda and os.environ.get('HNO_TRAIN_GRAPH_AUTOCAST', '1') != '0':
        # round 6: forward + loss + scaled backward of an autocast run replayed from a graph.  With our Adamax in device-stepped mode
        # GradScaler.step() / update() join the graph (no host read of the inf check: optim.Adamax._step_supports_amp_scaling);
        # otherwise (another optimizer, HNO_TRAIN_GRAPH_OPT=0) they stay eager behind the replay
        if os.environ.get('HNO_TRAIN_GRAPH_OPT', '1') != '0' and hasattr(optimizer, 'device_stepped') and not optimizer.is_device_stepped:
            entered_dev_opt = optimizer.device_stepped(scheduler)
        captured = CapturedStep(model, loss_fn, num_labels, label_mapping, data_parallel if world > 1 else None, optimizer=optimizer,
                                autocast=autocast, scaler=scaler)
    elif use_graph and not use_autocast and next(model.parameters()).is_cuda:
        # our Adamax moves its step counter, the learning rate and the per-batch cosine schedule onto the device, so that the update
        # is part of the captured step (HNO_TRAIN_GRAPH_OPT=0: keep optimizer and scheduler eager behind the replay)
        if os.environ.get('HNO_TRAIN_GRAPH_OPT', '1') != '0' and hasattr(optimizer, 'device_stepped') and not optimizer.is_device_stepped:
            entered_dev_opt = optimizer.device_stepped(scheduler)
        captured = CapturedStep(model, loss_fn, num_labels, label_mapping, data_parallel if world > 1 else None, optimizer=optimizer)
    dev_opt = getattr(optimizer, 'is_device_stepped', False)      # the tick kernel advances the schedule: no scheduler.step()

    for epoch in range(start_epoch, num_epochs):
        model.train()
        losses = []
        flow_it = iter(train_flow)
        nxt = next(flow_it, None)
        while nxt is not None:
            (x, y), nxt = nxt, None
            if captured is not None:
                loss = captured.step(x, y)
                if loss is not None:          # forward + loss + backward replayed; gradients (reduced over ranks) are in place
                    step_stats['replayed'] += 1
                    losses.append(loss.detach().clone())
                    if captured.steps_optimizer:         # the update (under autocast: GradScaler's step + update) was part of the replay
                        pass
                    elif scaler is not None:             # unscale, inf check (a host synchronisation), update, new scale
                        scaler.step(optimizer)
                        scaler.update()
                    else:
                        optimizer.step()
                    if scheduler is not None and not dev_opt:
                        scheduler.step()
                    nxt = next(flow_it, None)       # the host prepares the next batch and starts its PCIe copy while the GPU works
                    if nxt is not None:
                        captured.prefetch(*nxt)
                    continue
            step_stats['eager'] += 1
            x, y = x.to(device), y.to(device)
            y = labels_to_u8(y, num_labels, label_mapping)
            with autocast():
                with ops.expected_loss(y, loss_fn):
                    y_pred = model(x)
                loss = loss_fn(y_pred, y)
            losses.append(loss.detach())          # no host sync inside the step
            if data_parallel is not None:
                data_parallel.zero_grad()
            else:
                optimizer.zero_grad()
            if scaler is not None:
                scaler.scale(loss).backward()
                if data_parallel is not None:
                    data_parallel.allreduce_grads()
                scaler.step(optimizer)
                scaler.update()
            else:
                backward_from(loss)            # loss.backward() with a cached root gradient (no fill kernel per step)
                if data_parallel is not None:
                    data_parallel.allreduce_grads()
                optimizer.step()
            if scheduler is not None and not dev_opt:
                scheduler.step()
            # no reference to this step's autograd graph survives the iteration: a live loss tensor keeps the AccumulateGrad nodes
            # of the eager step alive, and a capture that follows then dies inside hipStreamEndCapture (DESIGN lesson 22)
            y_pred = loss = None
            nxt = next(flow_it, None)
        train_loss = mean_loss(losses)
        if dev_opt:
            optimizer.sync_from_device()       # step counters, lr and the scheduler's position back on the host objects
        log('', '-------------------------', f'Epoch: {epoch}', f'train_loss: {train_loss}')

        model.eval()
        losses = []
        with torch.no_grad():
            for x, y in valid_flow:
                x, y = x.to(device), y.to(device)
                y = labels_to_u8(y, num_labels, label_mapping)
                with autocast():
                    with ops.expected_loss(y, loss_fn):
                        y_pred = model(x)
                    losses.append(loss_fn(y_pred, y).detach())
        valid_loss = mean_loss(losses)
        if world > 1:               # same number on every rank: they must agree on best_epoch / min_loss
            t = torch.tensor([valid_loss if losses else 0.0, 1.0 if losses else 0.0], dtype=torch.float64,
                             device=data_parallel.device if dist.get_backend(data_parallel.group) == 'nccl' else 'cpu')
            dist.all_reduce(t, group=data_parallel.group)
            valid_loss = float(t[0] / t[1]) if float(t[1]) > 0 else float('nan')
        log(f'valid_loss: {valid_loss}')

        if (epoch + 1) % checkpoint_epoch == 0:
            if is_main:
                save_checkpoint(chkpt_path, epoch, model, optimizer, scheduler, min_loss, best_epoch, scaler)
            log('Standard checkpoint saved.')
        selection_epoch = int(num_epochs * selection_epoch_portion)
        if (epoch > selection_epoch or epoch == num_epochs - 1) and valid_loss < min_loss:
            min_loss, best_epoch = valid_loss, epoch
            if is_main:
                torch.save(model.state_dict(), model_path)
            if (epoch + 1) % checkpoint_epoch != 0:  # avoid saving twice
                if is_main:
                    save_checkpoint(chkpt_path, epoch, model, optimizer, scheduler, min_loss, best_epoch, scaler)
                log('Best checkpoint saved.')
    end_time = time.time()
    last_run.clear()
    last_run.update(device_stepped_optimizer=bool(dev_opt), schedules=dict(captured.schedule) if captured is not None else {})
    if entered_dev_opt:
        # the mode was entered for THIS run's captured steps: hand the caller's optimizer and scheduler back in torch's own stepping
        # form (a later scheduler.step() / eager loop would otherwise diverge from the device counters)
        captured = None
        optimizer.leave_device_stepped()

    if best_epoch is None and is_main:  # num_epochs == 0, i.e. no training
        torch.save(model.state_dict(), model_path)
    barrier()                       # model.pt is complete before any rank reads it
    if best_epoch is not None:
        model.load_state_dict(torch.load(model_path, weights_only=True, map_location=device))
    log('', f'Time used: {end_time - start_time:.2f} seconds.', f'Best epoch: {best_epoch}', f'Min loss: {min_loss}')
    return model


def testing(model, input_data, output_dir, label_mapping=None, output_origin=None, is_print=True, use_autocast=False,
            device=None, save_fn=None):
    """Prediction on the testing data with the reference's signature and protocol (experiments/train_test.py:332-426):
    batch size 1, `model.eval()`, per-sample wall time (first sample excluded from the average), label remapping,
    `prediction_time_memory.txt`.

    Changed on purpose: the class map is produced ON THE GPU (ops.label_output(): argmax fused into the upsampling
    kernel), so one uint8 volume crosses PCIe per sample instead of K float32 probability volumes followed by a host
    argmax (reference :398-408).  Writing NIfTI files needs SimpleITK (out of scope): predictions go
    through `save_fn(array, data_lists_test, index, images_dir, output_origin, suffix)` when given, else to
    `images/<index>_{true,pred}.npy`.  Returns (list of y_true or None, list of y_pred)."""
    from .. import ops
    from .utils import remap_labels
    import contextlib
    assert input_data.batch_size == 1
    os.makedirs(output_dir, exist_ok=True)
    images_dir = join(output_dir, 'images')
    os.makedirs(images_dir, exist_ok=True)
    test_num_batches = input_data.get_test_num_batches()
    data_lists_test = getattr(input_data, 'data_lists_test', None)
    if is_print:
        print('test_num_batches:', test_num_batches)
        print()
    test_flow = input_data.get_test_flow()
    model.to(device)
    model.eval()
    if is_print:
        print('Testing started')
        print(output_dir)

    def save(arr, i, suffix):
        if save_fn is not None:
            save_fn(arr, data_lists_test, i, images_dir, output_origin, suffix)
        else:
            np.save(join(images_dir, f'{i}{suffix}.npy'), arr)

    start_time = time.time()
    predict_times, y_trues, y_preds = [], [], []
    for i, xy in enumerate(test_flow):
        s_time = time.time()
        y_true = None
        if isinstance(xy, (tuple, list)):
            x, y = xy
            y_true = np.asarray(y, dtype=np.uint8)[0, 0]  # (1, 1, D, H, W) to (D, H, W)
        else:
            x = xy
        x = x.to(device)
        with torch.no_grad(), ops.label_output(), (torch.autocast('cuda', dtype=torch.bfloat16) if use_autocast else contextlib.nullcontext()):
            yp = model(x)                                   # (1, 1, D, H, W) uint8 labels
        y_pred = np.asarray(yp.detach().to('cpu'))[0, 0]    # the .to('cpu') is the synchronisation point
        e_time = time.time()
        if y_true is not None:
            save(y_true, i, '_true')
        if label_mapping is not None:                       # outside the timed span, as in the reference (:409-410)
            y_pred = np.asarray(remap_labels(yp, label_mapping).to('cpu'))[0, 0].astype(np.uint8)
        save(y_pred, i, '_pred')
        y_trues.append(y_true)
        y_preds.append(y_pred)
        if i != 0:  # Skip the first iteration which involves model initialization
            predict_times.append(e_time - s_time)
    end_time = time.time()
    avg = float(np.mean(predict_times)) if predict_times else float('nan')
    on_gpu = torch.cuda.is_available() and device is not None and torch.device(device).type == 'cuda'
    lines = [f'Average prediction time: {avg}']
    if on_gpu:
        lines += [f'max_memory_reserved: {torch.cuda.max_memory_reserved(device) / 1024 ** 2:.2f} MiB',
                  f'max_memory_allocated: {torch.cuda.max_memory_allocated(device) / 1024 ** 2:.2f} MiB']
    if is_print:
        print(f'\nTime used: {end_time - start_time:.2f} seconds.')
        print('\n'.join(lines))
    with open(join(output_dir, 'prediction_time_memory.txt'), 'w') as f:
        print('\n'.join(lines), file=f)
    return y_trues, y_preds


def save_checkpoint(chkpt_path, epoch, model, optimizer, scheduler, min_loss, best_epoch, scaler):
    """Same dict keys as the reference (train_test.py:262-273)."""
    checkpoint = {
        'epoch': epoch,
        'model_state_dict': model.state_dict(),
        'optimizer_state_dict': optimizer.state_dict(),
        'scheduler_state_dict': scheduler.state_dict() if scheduler is not None else None,
        'min_loss': min_loss,
        'best_epoch': best_epoch,
    }
    if scaler is not None:
        checkpoint['scaler_state_dict'] = scaler.state_dict()
    torch.save(checkpoint, chkpt_path)


def load_checkpoint(chkpt_path, model, optimizer, scheduler, scaler, device):
    """Reference train_test.py:276-286."""
    checkpoint = torch.load(chkpt_path, weights_only=False, map_location=device)
    model.load_state_dict(checkpoint['model_state_dict'])
    optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
    if scheduler is not None and checkpoint.get('scheduler_state_dict') is not None:
        scheduler.load_state_dict(checkpoint['scheduler_state_dict'])
    if scaler is not None:
        scaler.load_state_dict(checkpoint['scaler_state_dict'])
    return checkpoint['epoch'], checkpoint['min_loss'], checkpoint['best_epoch']


def get_losses_from_file(filename):
    """Parses `train_loss:` / `valid_loss:` lines (reference train_test.py:289-302)."""
    train_loss, valid_loss = [], []
    with open(filename) as f:
        for ln in f:
            if 'train_loss' in ln:
                train_loss.append(float(re.findall('train_loss: (.+)', ln)[0]))
            elif 'valid_loss' in ln:
                valid_loss.append(float(re.findall('valid_loss: (.+)', ln)[0]))
    assert len(train_loss) == len(valid_loss)
    return train_loss, valid_loss
