#!/usr/bin/env python3
"""Benchmark of the accelerated hot path (contract in the task prompt / BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / [4]): HNOSeg-XS, BraTS'23 config (filters 24, 8 blocks x 3
frequency-domain convs, modes (10,14,14)), per-GPU batch 2 of synthetic 4-modal 128^3 fp32
volumes, random-init weights.  One step = forward + PCC loss + backward + gradient all-reduce
(N > 1) + Adamax update.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (by total HIP-event
time inside the timed region); `cpu_baseline` times the CPU oracle (the torch-CPU port of the
reference op sequence) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL_CFG = dict(in_channels=4, out_channels=4, filters=24, num_transform_blocks=[3] * 8, num_modes=(10, 14, 14))
VOL = (128, 128, 128)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
ALGO_BYTES_PER_VOLUME = 4.67e9  # fwd 1.596 GB + bwd 3.073 GB


def effective_cpus():
    """CPUs this process may really use: affinity mask and cgroup quota, not os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(budget_s=40.0):
    """CPU oracle (kind "port"): fwd + PCC + bwd + Adamax of HNOSeg-XS on the host cores, on a
    bounded sample of the benchmark workload (one 4x128^3 volume per step when it fits the time
    budget, otherwise one 4x64^3 volume, which is said in `sample`)."""
    from oracle import hno_oracle as O
    import multimodal_3d_image_segmentation_amd as pkg
    ncores = min(effective_cpus(), 64)
    torch.set_num_threads(ncores)
    torch.manual_seed(0)
    model = pkg.nets.HNOSegXS(**MODEL_CFG)   # parameters only; the oracle does the math
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adamax(list(params.values()), lr=5e-3)
    g = torch.Generator().manual_seed(1234)

    def make(n):
        return (torch.randn((1, 4, n, n, n), generator=g), torch.randint(0, 4, (1, 1, n, n, n), generator=g).float())

    def step(xx, ll):
        y = O.hnosegxs_forward(params, xx, MODEL_CFG['num_transform_blocks'], MODEL_CFG['num_modes'])
        loss = O.pcc_loss(y, O.to_categorical(ll, 4))
        opt.zero_grad()
        loss.backward()
        opt.step()
        return float(loss)

    x64, l64 = make(64)
    step(x64, l64)                      # warm-up
    t0 = time.time()
    step(x64, l64)
    t64 = time.time() - t0
    # a 128^3 step costs ~9-12x a 64^3 one: aim at ~12 s of CPU work (at least one step), never above the budget
    est128 = 10.0 * t64
    if est128 < budget_s:
        n, steps = 128, max(1, min(8, int(12.0 / max(est128, 1e-3))))
    else:
        n, steps = 64, max(1, int(budget_s / 4 / max(t64, 1e-3)))
    xs, ls = (make(128) if n == 128 else (x64, l64))
    t0 = time.time()
    for _ in range(steps):
        step(xs, ls)
    dt = time.time() - t0
    return {'value': steps / dt, 'unit': 'volumes/s', 'cores': ncores, 'kind': 'port',
            'sample': f'{steps} step(s) of one synthetic 4x{n}^3 volume, fwd+PCC+bwd+Adamax, oracle/hno_oracle.py '
                      f'(torch-CPU fp32 port of the reference op sequence), {dt:.1f} s; 64^3 probe step {t64:.2f} s'}


def launch_ranks(n, dry_run=False):
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, the same command line);
    rank 0's stdout -- the JSON line -- passes through, the other ranks' stdout is dropped.  Returns the worst exit code.
    dry_run: the children only initialise a gloo group and all-reduce a flat buffer (no GPU needed: the CPU suite runs this)."""
    import socket
    import subprocess
    if not dry_run:
        have = torch.cuda.device_count()       # does not initialise the GPU runtime
        if have < n:
            print(f'[bench] --gpus {n} requested but only {have} GPU(s) are visible', file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def dry_run_rank(args):
    """`--dry-run`: what a rank of `--gpus N` does up to and including its first collective, without a GPU -- read RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the environment, join a gloo group, all-reduce a flat gradient-sized buffer (HNOSeg-XS: 28 248 floats),
    check the mean, rank 0 prints one JSON line.  `--dry-run-fail-rank R` makes rank R exit with code 3 (exit-code propagation)."""
    import torch.distributed as dist
    world, rank, local_rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK']), int(os.environ['LOCAL_RANK'])
    assert os.environ.get('MASTER_ADDR') and os.environ.get('MASTER_PORT')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with {args.gpus} ranks')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    flat = torch.full((28248,), float(rank + 1))
    dist.all_reduce(flat)
    flat /= world
    ok = bool(torch.all(flat == (world + 1) / 2.0))
    print('this line must not reach the launcher\'s stdout from ranks > 0' if rank else
          json.dumps({'dry_run': True, 'rccl_ranks': dist.get_world_size(), 'backend': 'gloo', 'rank': rank, 'local_rank': local_rank,
                      'master_port': int(os.environ['MASTER_PORT']), 'allreduce_ok': ok}), flush=True)
    dist.destroy_process_group()
    if args.dry_run_fail_rank == rank:
        raise SystemExit(3)
    raise SystemExit(0 if ok else 4)


def probe_store_path():
    """a rendezvous file every rank of THIS launch derives alone (ranks cannot talk before they have a group): the launcher's pid (the
    parent of every RANK: torchrun's agent, or launch_ranks) and the master port are common to all ranks of one launch and differ between
    launches.  Computed by the rank and handed to its probe child in HNO_PROBE_STORE (the child's own parent is the rank)."""
    import tempfile
    if os.environ.get('HNO_PROBE_STORE'):
        return os.environ['HNO_PROBE_STORE']
    return os.path.join(tempfile.gettempdir(), f"hno_probe_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}")


def probe_captured_allreduce_child():
    """`--probe-capture` (a child of every rank, started BEFORE the rank touches the GPU): its own RCCL group over a file store, an
    all-reduce of a gradient-sized flat buffer CAPTURED into a HIP graph, three replays with changing inputs, results checked.  Exit 0 =
    a collective inside a graph works across these ranks on this stack.  A mismatch inside a captured collective hangs instead of
    raising, so this runs in a throw-away process under the parent's timeout, never in the benchmark process itself."""
    import torch.distributed as dist
    world, rank, local_rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK']), int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist.init_process_group('nccl', init_method='file://' + probe_store_path(), rank=rank, world_size=world, device_id=dev)
    flat = torch.zeros(28248, device=dev)
    src = torch.zeros(28248, device=dev)
    dist.all_reduce(flat)                       # eager once: communicator set-up is not capturable
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
            flat.copy_(src)
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    torch.cuda.current_stream().wait_stream(side)
    ok = True
    for it in range(3):
        src.fill_(float(rank + 1 + it))
        graph.replay()
        torch.cuda.synchronize()
        ok = ok and bool(torch.all(flat == (world + 1) / 2.0 + it))
    dist.destroy_process_group()
    raise SystemExit(0 if ok else 5)


def probe_captured_allreduce(timeout_s=150.0):
    """-> True when the throw-away probe (above) succeeded on THIS rank within the timeout; the ranks then agree on the minimum."""
    import subprocess
    try:
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--probe-capture'], stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL, env=dict(os.environ, HNO_PROBE_STORE=probe_store_path()))
    except OSError:
        return False
    try:
        return child.wait(timeout=timeout_s) == 0
    except subprocess.TimeoutExpired:
        child.kill()                            # exactly the process we started
        child.wait()
        return False


def secondary_configs(pkg, dev):
    """BASELINE cfg3 / cfg4 and the inference protocol, timed by the same process after the headline (N = 1 only; each wrapped so
    that a failure cannot take the headline line down).  ms per fwd + PCC + bwd step, HIP-graph replay, synthetic data."""
    import contextlib
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    nets = pkg.nets
    out = {}

    def time_step(ctor, shape, bf16, replays=10):
        torch.manual_seed(0)
        model = ctor().to(dev)
        x = torch.randn(shape, device=dev)
        lab = pkg.ops.labels_prepare(torch.randint(0, 4, (shape[0], 1) + shape[2:], device=dev).float(), 4)
        loss_fn = custom_losses.PCCLoss()
        ac = (lambda: torch.autocast('cuda', dtype=torch.bfloat16)) if bf16 else contextlib.nullcontext

        def step():
            for p in model.parameters():
                p.grad = None
            with ac():
                with pkg.ops.expected_loss(lab, loss_fn):
                    y = model(x)
                loss = loss_fn(y, lab)
            loss.backward()
        step(); step()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                step()
        torch.cuda.current_stream().wait_stream(side)
        gr.replay()
        torch.cuda.synchronize()
        ms = float('inf')
        for _ in range(3):       # the best of three bursts (one burst of ten once read 10.6 ms for a 7.6 ms step on a box of the pool)
            t0 = time.perf_counter()
            for _ in range(replays):
                gr.replay()
            torch.cuda.synchronize()
            ms = min(ms, (time.perf_counter() - t0) / replays * 1e3)
        # the step's ALGORITHMIC bytes: what every launch of one eager step declares to the profiler (DESIGN section 4's per-kernel
        # definitions; launches that declare none count as zero, so the sum is a lower bound)
        algo = None
        try:
            if os.environ.get('HNO_BENCH_SEC_NOALGO'):
                raise RuntimeError('skipped')
            with pkg._lib.KernelProfile(max_records=4096) as kp:
                step()
            torch.cuda.synchronize()
            algo = sum(nb for _, _, nb in kp.records)
        except Exception:   # noqa: BLE001
            pass
        del gr, model, x
        torch.cuda.empty_cache()
        return round(ms, 3), algo
    cases = {
        'cfg3_fnoseg_2x4x128^3_bf16_autocast_ms_per_step': (lambda: nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier'), (2, 4, 128, 128, 128), True),
        'cfg3_fnoseg_2x4x128^3_fp32_ms_per_step': (lambda: nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier'), (2, 4, 128, 128, 128), False),
        'cfg4_vnetds_1x4x160x192x128_bf16_autocast_ms_per_step': (lambda: nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4]), (1, 4, 160, 192, 128), True),
        'cfg4_vnetds_1x4x160x192x128_fp32_ms_per_step': (lambda: nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4]), (1, 4, 160, 192, 128), False),
        # the attention family (reference tensorflow/experiments/config_files/config_hartleymha.ini:24-29: 12 filters, 16 blocks, 4 heads, modes 10-14-14, patch 2 x 2 x 2)
        'hartleymha_1x4x128^3_fp32_ms_per_step': (lambda: nets.HartleyMHASeg(4, 4, 12, 16, 4, (10, 14, 14), (2, 2, 2)), (1, 4, 128, 128, 128), False),
    }
    algo_bytes = {}
    for key, (ctor, shape, bf16) in cases.items():
        try:
            out[key], algo_bytes[key] = time_step(ctor, shape, bf16)
        except Exception as exc:   # noqa: BLE001
            out[key] = f'failed: {exc!r}'[:200]
    # rooflines of the two other BASELINE configurations (round 5; the headline's is the top-level `roofline`):
    # cfg3 FNOSeg is bandwidth-bound: achieved = algorithmic bytes of the step (sum over its launches, as declared to the profiler:
    #   DESIGN section 4) / replayed step time, against the 8 TB/s HBM peak;
    # cfg4 V-Net-DS is matrix-bound: 463 GFLOP per volume fwd + bwd (154.3 GFLOP forward [probed] x 3) / step time,
    #   against the 2.5 PFLOP/s dense bf16 MFMA peak of MI355X_MICROARCH.md (fp32 runs: the 157 TFLOP/s fp32 matrix peak).
    for tag, key in (('cfg3_roofline', 'cfg3_fnoseg_2x4x128^3_bf16_autocast_ms_per_step'), ('cfg3_fp32_roofline', 'cfg3_fnoseg_2x4x128^3_fp32_ms_per_step')):
        ms, nb = out.get(key), algo_bytes.get(key)
        if isinstance(ms, float) and nb:
            ach = nb / (ms * 1e-3) / 1e9
            out[tag] = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                        'algorithmic_bytes_per_step': nb}
    for tag, key, peak in (('cfg4_roofline', 'cfg4_vnetds_1x4x160x192x128_bf16_autocast_ms_per_step', 2500.0),
                           ('cfg4_fp32_roofline', 'cfg4_vnetds_1x4x160x192x128_fp32_ms_per_step', 157.0)):
        ms = out.get(key)
        if isinstance(ms, float):
            ach = 0.463 / (ms * 1e-3)       # TFLOP/s
            out[tag] = {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                        'flops_per_step': 463e9}
    # the reference's published metric: single-image inference at 240 x 240 x 155 incl. host copies (README.md:10: V100 < 0.24 s), and the
    # same volume in the array order its loader produces ((z, y, x) = (155, 240, 240), experiments/utils.py:270: 78 planes of 121 x 121)
    for size in ((240, 240, 155), (155, 240, 240)):
        key = 'hnosegxs_inference_%dx%dx%d' % size
        try:
            model = nets.HNOSegXS(**MODEL_CFG).to(dev).eval()
            xh = torch.randn((1, 4) + size).pin_memory()
            ts, gs = [], []
            for i in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                xd = xh.to(dev, non_blocking=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                with torch.no_grad(), pkg.ops.label_output():
                    yp = model(xd)
                e1.record()
                _ = yp.to('cpu')
                if i:
                    ts.append(time.perf_counter() - t0)
                    gs.append(e0.elapsed_time(e1) * 1e-3)
            out[key + '_seconds_per_image'] = round(sum(ts) / len(ts), 5)
            out[key + '_gpu_forward_seconds'] = round(sum(gs) / len(gs), 5)   # the forward pass alone (item plane kernels)
        except Exception as exc:   # noqa: BLE001
            out[key + '_seconds_per_image'] = f'failed: {exc!r}'[:200]
    return out


def dp_path_secondary(headline_ms):
    """The full-size cfg2 step through the N > 1 code path on ONE rank (a fresh child process: its own RCCL group of one rank,
    `--dp-path`), next to the headline: ms per step, the ratio, and the host time the step's launches take."""
    import subprocess
    out = {}
    # two forms: the default (replay + one eager flat all-reduce + one eager Adamax launch) and, round 4, the whole step as ONE graph
    # replay (HNO_DP_CAPTURE_ALLREDUCE=1: the collective and the device-stepped Adamax are nodes of the graph)
    for key, env in (('dp_path_1rank', {}), ('dp_path_1rank_one_replay', {'HNO_DP_CAPTURE_ALLREDUCE': '1'})):
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), '--dp-path', '--steps', '30', '--warmup', '5', '--no-secondary',
                                  '--no-cpu-baseline', '--no-kernel-profile'], capture_output=True, text=True, timeout=600,
                                 env=dict(os.environ, **env))
            line = [ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1]
            d = json.loads(line)
            out[f'{key}_ms_per_step'] = d['ms_per_step']
            out[f'{key}_over_headline'] = round(d['ms_per_step'] / headline_ms, 4)
            out[f'{key}_host_us_per_step'] = d['config'].get('host_us_per_step')
            out[f'{key}_launch'] = d['config'].get('launch')
        except Exception as exc:
            out[f'{key}_ms_per_step'] = f'failed: {exc!r}'[:200]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=2, help='per-GPU batch (BASELINE config: 2)')
    ap.add_argument('--bursts', type=int, default=5, help='the timed region of --steps steps is repeated this often; the median burst is reported')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying a HIP graph')
    ap.add_argument('--no-secondary', action='store_true', help='skip the cfg3 / cfg4 / inference timings appended to the line')
    ap.add_argument('--dp-path', action='store_true',
                    help='one rank, but through the data-parallel step as N > 1 ranks run it: RCCL group of one rank, flat gradient '
                         'buffer written by the backward kernels, bucket all-reduces on the communication stream behind the graph '
                         'replay (what `secondary.dp_path_1rank_ms_per_step` reports)')
    ap.add_argument('--dry-run', action='store_true',
                    help='rank plumbing only, no GPU: every rank joins a gloo group and all-reduces a flat buffer (the CPU suite runs '
                         '`--gpus 2 --dry-run` through launch_ranks)')
    ap.add_argument('--dry-run-fail-rank', type=int, default=-1, help='with --dry-run: this rank exits with code 3')
    ap.add_argument('--probe-capture', action='store_true', help='internal: the throw-away child that tries an RCCL all-reduce inside a HIP graph')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.probe_capture:
        probe_captured_allreduce_child()
    if args.gpus > 1 and world == 1 and 'RANK' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  This process has not touched the GPU yet (nothing
        # above initialises HIP) and never will: the ranks are fresh child processes, this one only waits for them.
        raise SystemExit(launch_ranks(args.gpus, dry_run=args.dry_run))
    if args.dry_run:
        dry_run_rank(args)
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with {args.gpus} ranks')
    # N > 1: whether the gradient all-reduce (and with it the Adamax update) can be part of the replayed graph is TRIED, in a throw-away
    # child of every rank, before this process touches the GPU (a captured collective that mismatches hangs; nothing can be retried in
    # here).  HNO_DP_CAPTURE_ALLREDUCE=1 / 0 skip the probe and force / forbid the form.
    env_cap = os.environ.get('HNO_DP_CAPTURE_ALLREDUCE', '')
    probe_ok = None
    if world > 1 and env_cap not in ('0', '1') and not args.no_graph:
        probe_ok = probe_captured_allreduce()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    elif args.dp_path:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    distributed = world > 1 or args.dp_path

    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    from multimodal_3d_image_segmentation_amd.parallel import FlatGradReplica

    torch.manual_seed(0)
    model = pkg.nets.HNOSegXS(**MODEL_CFG).to(dev)
    rep = FlatGradReplica(model, force_distributed=args.dp_path and world == 1)
    opt = pkg.optim.Adamax(model.parameters(), lr=5e-3)   # one-launch multi-tensor Adamax (hno_adamax_multi)
    loss_fn = custom_losses.PCCLoss()
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((B, 4) + VOL, device=dev, generator=g)
    labels = torch.randint(0, 4, (B, 1) + VOL, device=dev, generator=g).float()
    # one batched weight-gradient slab reduction per backward (plain leaf parameters here).  Not in eager data-parallel steps: the
    # buckets leave from hooks DURING backward, a reduction deferred to its end would land after its bucket was sent
    pkg.ops.set_defer_reduce(not distributed)

    def fwd_bwd():
        # the label conversion is part of every step, as in the reference loop (to_categorical, train_test.py:150-152);
        # here it yields the uint8 class map the loss kernels read (the one-hot tensor never exists)
        lab_u8 = pkg.ops.labels_prepare(labels, 4)
        with pkg.ops.expected_loss(lab_u8, loss_fn):     # as training() runs the step: the head takes the loss sums in its own pass
            y = model(x)
        loss = loss_fn(y, lab_u8)
        rep.zero_grad()
        pkg.ops.backward_from(loss)      # loss.backward() with a cached root gradient: autograd's ones_like fill is not part of the model
        return loss


    def eager_step():
        loss = fwd_bwd()
        rep.allreduce_grads()
        opt.step()
        return loss

    for _ in range(max(args.warmup, 2)):   # also creates the twiddle tables / kernel attributes (not capturable)
        eager_step()

    # captured steps: the two halves of the batch MAY run as two concurrent passes on two streams of the graph (experiments.train_test.
    # SampleSplit) -- decided as training()'s CapturedStep decides it: both forms of forward + loss + backward are captured, replayed
    # three times each, and the faster one is kept (train_test.choose_schedule; HNO_SPLIT_STREAMS=1 / 0 force the answer)
    from multimodal_3d_image_segmentation_amd.experiments.train_test import SampleSplit, choose_schedule
    split_mode = False if args.no_graph else SampleSplit.candidate(model, loss_fn, x)
    split = SampleSplit(model) if split_mode else None
    schedule_ms = None
    if split is not None:
        with torch.no_grad():
            model(x[:B // 2])                 # tables / kernel attributes of the half-batch shapes exist before the capture

    def fwd_bwd_split():
        lab_u8 = pkg.ops.labels_prepare(labels, 4)
        return split.fwd_bwd(x, lab_u8, loss_fn, zero_grad=rep.zero_grad)

    if split_mode == 'measure':
        prev_defer = pkg.ops.set_defer_reduce(True)
        rep.set_hooks_enabled(False)
        try:
            use, ms_one, ms_split = choose_schedule(lambda: fwd_bwd().detach(), fwd_bwd_split,
                                                    agree=rep.all_ranks_ok if distributed else None, what=' of bench.py')
            schedule_ms = {'one_pass': round(ms_one, 3), 'two_streams': round(ms_split, 3)}
        except Exception as exc:   # noqa: BLE001
            print(f'[bench] schedule measurement failed ({exc!r}); one pass over the batch', file=sys.stderr)
            use = False
        pkg.ops.set_defer_reduce(prev_defer)
        rep.set_hooks_enabled(True)
        rep.zero_grad()
        for q in split.tparams:
            q.grad = None
        if not use:
            split = None

    def fwd_bwd_captured():
        return fwd_bwd() if split is None else fwd_bwd_split()

    # forward + loss + backward are captured ONCE into a HIP graph and replayed: ~120 kernel launches per step cost no host
    # time, so the GPU is never launch-bound.  Adamax stays an eager launch behind the replay.
    # Data-parallel runs replay the same graph: the backward kernels write their weight gradients straight into the flat
    # gradient buffer (the few that cannot are copied there by kernels inside the graph, rep.finish_capture()), and the bucket
    # gradient all-reduce is launched from the compute stream behind the replay (rep.allreduce_flat(): one RCCL launch for
    # HNOSeg-XS's 113 KB, no Python per parameter).  Round 2 ran N > 1 eagerly with ~60 Python hooks per backward; the hooks
    # remain the path of `training()` for models whose gradients are worth overlapping (V-Net-DS: 90 MB).
    graph = None
    # HNO_DP_CAPTURE_ALLREDUCE=1: also capture the gradient all-reduce into the graph (measured on one rank only; the default keeps it an
    # eager launch behind the replay because a capture of RCCL collectives across several GPUs could not be tested from here)
    # N > 1 (round 5): the one-replay form when the probe above succeeded on EVERY rank (rep.all_ranks_ok: a MIN all-reduce of the
    # flags), else the eager collective behind the replay; --dp-path on one rank keeps the measured default (eager) unless asked
    if env_cap in ('0', '1'):
        capture_allreduce = distributed and env_cap == '1'
    elif world > 1:
        capture_allreduce = bool(rep.all_ranks_ok(bool(probe_ok)))
    else:
        capture_allreduce = False
    # round 4: Adamax with its step counter / learning rate on the device (optim.Adamax.device_stepped): the update has no per-step host
    # argument and is captured behind backward -- a step is ONE graph replay.  With replicas it has to follow the all-reduce, so it is
    # part of the graph only when the collective is (HNO_DP_CAPTURE_ALLREDUCE=1); otherwise it stays one eager launch behind it.
    opt_in_graph = os.environ.get('HNO_BENCH_OPT_IN_GRAPH', '1') == '1' and opt.device_stepped(None) and (not distributed or capture_allreduce)
    if not args.no_graph:
        try:
            torch.cuda.synchronize()
            rep.set_hooks_enabled(False)
            pkg.ops.set_defer_reduce(True)     # inside a captured step nothing is sent before backward ends: batch the reductions
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                # thread_local: the RCCL watchdog thread polls events while we capture; it must not invalidate the capture
                with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                    static_loss = fwd_bwd_captured()
                    if distributed:
                        rep.finish_capture()
                        if capture_allreduce:      # the collective becomes a node of the graph (RCCL kernels are capturable)
                            rep.allreduce_flat()
                    if opt_in_graph:
                        opt.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if opt_in_graph:
                opt.finish_capture()           # the chunk table built during the capture -> device, once (not a copy node per replay)
        except Exception as exc:   # capture unsupported on this stack: run eagerly and say so
            print(f'[bench] HIP graph capture failed ({exc!r}); running eagerly', file=sys.stderr)
            graph = None
            rep.set_hooks_enabled(True)
            if distributed:
                pkg.ops.set_defer_reduce(False)
        if world > 1 and not rep.all_ranks_ok(graph is not None) and graph is not None:
            # a capture that failed on ANY rank is dropped on EVERY rank: a replaying rank sends one flat all-reduce (possibly inside its
            # graph), an eager rank one per bucket -- mixed, the collectives no longer match and the job hangs
            print('[bench] HIP graph capture failed on another rank; running eagerly', file=sys.stderr)
            graph = None
            rep.set_hooks_enabled(True)
            pkg.ops.set_defer_reduce(False)
        if graph is None and opt_in_graph:
            opt_in_graph = False              # (the update then runs as an eager launch behind backward; still device-stepped)

    host_s = [0.0]

    def step():
        if graph is None:
            return eager_step()
        t_h = time.perf_counter()
        graph.replay()
        if distributed and not capture_allreduce:
            rep.allreduce_flat()
        if not opt_in_graph:
            opt.step()
        host_s[0] += time.perf_counter() - t_h
        return static_loss

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region is EXACTLY --steps steps between two fences (barrier + synchronize).  It is repeated --bursts times back to back
    # and the MEDIAN burst is reported (verdict round 5, item 7: one 45 ms burst on a pool whose boxes and clocks wander is a coin
    # toss); every burst's time is in config.ms_per_step_bursts, so the single-burst number the contract describes is there too
    # (the first entry).  For N > 1 each burst's time is the MAX over ranks, taken before the median.
    burst_dt = []
    host_s[0] = 0.0
    for _ in range(max(1, args.bursts)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        fence()
        burst_dt.append(time.perf_counter() - t0)
    if world > 1:
        tmax = torch.tensor(burst_dt, device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        burst_dt = [float(v) for v in tmax.tolist()]
    dt = sorted(burst_dt)[(len(burst_dt) - 1) // 2]       # median (the lower one of an even count)
    host_us_per_step = host_s[0] / (args.steps * len(burst_dt)) * 1e6 if graph is not None else None
    # per-kernel HIP-event durations: a few eager steps of the same workload right after the timed
    # region (events cannot be recorded inside a graph replay)
    prof = None
    prof_steps = min(args.steps, 5)
    if not args.no_kernel_profile:
        prof = pkg._lib.KernelProfile(max_records=400 * prof_steps + 64)
        with prof:
            for _ in range(prof_steps):
                eager_step()
        fence()
    value = world * B * args.steps / dt

    if rank == 0:
        roofline = None
        kernels = {}
        if prof:
            summ = prof.summary()
            kernels = {k: {'calls_per_step': c / prof_steps, 'avg_us': round(avg * 1e3, 2),
                           'total_ms_per_step': round(s / prof_steps, 4),
                           'algorithmic_GBps': round(nb / (s * 1e-3) / 1e9, 1) if nb else None}
                       for k, (c, s, avg, nb) in sorted(summ.items(), key=lambda kv: -kv[1][1])}
            dom = max(summ.items(), key=lambda kv: kv[1][1])[0]
            calls, tot_ms, avg_ms, tot_bytes = summ[dom]
            ab = tot_bytes / calls          # mean ALGORITHMIC bytes per launch (recorded by the launcher)
            if ab:
                achieved = tot_bytes / (tot_ms * 1e-3) / 1e9
                traffic = None
                tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
                if os.path.exists(tpath):
                    tj = json.load(open(tpath))      # PMC passes of the same command (tools/make_hbm_traffic.py; profiles/README.md)
                    traffic = tj.get('per_kernel_bytes_per_launch', tj).get(dom)
                roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                            'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                            # PMC counters need rocprofv3 around the process: the number is the committed PMC pass of this command
                            'traffic_source': 'profiles/hbm_traffic.json (static: rocprofv3 --pmc passes of this command, tools/make_hbm_traffic.py)' if traffic is not None else None,
                            'algorithmic_bytes_per_launch': ab, 'avg_launch_us': round(avg_ms * 1e3, 2)}
        out = {
            'metric': 'volumes/sec fwd+bwd, HNOSeg-XS 4-modal 128^3',
            'value': round(value, 3), 'unit': 'volumes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': "HNOSeg-XS BraTS'23 config (filters 24, 8 blocks x 3, modes 10-14-14), "
                                   "synthetic 4-modal 128^3 fp32, step = fwd + PCC loss + bwd + grad all-reduce + Adamax",
                       'per_gpu_batch': B, 'global_batch': B * world, 'parallelism': f'dp{world}',
                       'launch': ('hip-graph replay (fwd+loss+bwd)' + ((' + one flat gradient all-reduce ' + ('inside the graph' if capture_allreduce else 'behind the replay')) if distributed else '')
                                  + (' + Adamax inside the graph (device-side step counter): one replay per step' if opt_in_graph else ' + eager Adamax')) if graph is not None else
                                 ('eager; gradient buckets all-reduced from backward hooks on a comm stream' if distributed else 'eager'),
                       'grad_buckets': len(rep.buckets) if distributed else 0,
                       # ranks the process group really has (the driver can check that N ranks were seen) and how the collective form was chosen
                       'rccl_ranks': dist.get_world_size() if distributed else 1,
                       'allreduce_in_graph': bool(capture_allreduce),
                       'allreduce_in_graph_chosen_by': ('env HNO_DP_CAPTURE_ALLREDUCE' if env_cap in ('0', '1') else
                                                        (f'probe on every rank (this rank: {probe_ok})' if world > 1 else 'n/a')),
                       'schedule_measured_ms': schedule_ms,
                       # every timed burst of --steps steps, in order; ms_per_step / value are the median burst
                       'ms_per_step_bursts': [round(v / args.steps * 1e3, 4) for v in burst_dt],
                       'ms_per_step_min_max': [round(min(burst_dt) / args.steps * 1e3, 4), round(max(burst_dt) / args.steps * 1e3, 4)],
                       'dp_path_on_one_rank': bool(args.dp_path and world == 1),
                       'host_us_per_step': None if host_us_per_step is None else round(host_us_per_step, 1),
                       # the captured step's schedule: the batch's two halves as concurrent passes on two streams of the graph
                       # (experiments.train_test.SampleSplit; HNO_SPLIT_STREAMS=0: one pass over the batch)
                       'schedule': ('two half-batches on two streams of one graph' if (graph is not None and split is not None) else 'one pass over the batch'),
                       'final_loss': round(float(loss), 6)},
            'roofline': roofline,
            'whole_step_roofline': {'algorithmic_GB_per_volume': ALGO_BYTES_PER_VOLUME / 1e9,
                                    'hbm_bound_volumes_per_s_per_gpu': round(HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_VOLUME, 1),
                                    'frac': round(value / world / (HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_VOLUME), 4)},
            'kernels': kernels,
        }
        if world == 1 and not args.no_secondary and not args.dp_path:
            del graph
            out['secondary'] = secondary_configs(pkg, dev)
            out['secondary'].update(dp_path_secondary(out['ms_per_step']))
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1 or args.dp_path:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
