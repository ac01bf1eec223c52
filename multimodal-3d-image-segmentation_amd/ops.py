"""Functional + autograd wrappers over the libhno C ABI (include/hno.h).

Each ``torch.autograd.Function`` here is one fused op of the hot path; torch tensors are only
containers (allocation, lifetime, autograd graph).  Nothing in this file computes on the CPU
or through ATen math kernels: if libhno.so is missing or the tensors are not on a GPU the
call raises.
"""
import contextlib
import os
import threading

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

ACT_NONE, ACT_SELU, ACT_ELU = 0, 1, 2
ACT_BF16 = 0x1000   # ORed into the `act` argument of the pointwise entry points: bf16 matrix-core arithmetic (autocast)
ACT_IO16 = 0x2000   # with ACT_BF16: the block's input / output (and the output's gradient) are bf16 tensors IN MEMORY (include/hno.h)
ACT_SIGMOID = 3     # elementwise ActFn only (model output activation); never passed to the fused conv / transform epilogues
_ACT_IDS = {None: ACT_NONE, 'none': ACT_NONE, 'selu': ACT_SELU, 'elu': ACT_ELU}
LOSS_KINDS = {'pcc': 0, 'dice': 1, 'expdice': 2}


def act_id(act):
    if callable(act):
        act = getattr(act, '__name__', act)
    if act not in _ACT_IDS:
        raise ValueError(f'activation {act!r} is not supported by the HIP path (selu, elu or None)')
    return _ACT_IDS[act]


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.HnoError('the HIP path needs CUDA/HIP tensors; there is no CPU fallback '
                                '(use oracle/ for CPU checks)')


def _f32c(t):
    """fp32, contiguous (a channel-padded activation is repacked: chan_stride / to_layout below)"""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() == 5 and not t.is_contiguous() and chan_stride(t) is not None:
        return to_layout(t, None)
    return t.contiguous()


# ------------------------------------------------------------------------- channel-padded activations
# HNOSeg-XS works on 65^3 volumes: V = 274625 floats per channel is odd, so in a contiguous (B, C, V) tensor every channel row of
# every tile starts at a different offset inside a 32-byte HBM sector and the pointwise kernels fetch 1.25x - 1.33x the bytes they
# use (profiles/r03_a_pmc_calibration_planar_copy.json, tools/dbg/pw_padded_stride.py).  Inside the HNOSeg-XS path activations
# therefore live with the channel stride rounded up to a multiple of 32 floats (128 B):
#     strides (C ld, ld, H W, W, 1),  ld = round_up(V, 32)
# * the pointwise kernels are called unchanged with V := ld (they never look at spatial coordinates); the transforms and the stem
#   take the stride as an argument (hno_*_ld, ldbc / ldy);
# * invariants: forward padding is FINITE (zero out of the stem, the inverse transform and to_layout; act(W pad + b) after a
#   pointwise layer), gradient padding is exactly ZERO (to_layout and the inverse transform zero it, the pointwise backward maps
#   zeros to zeros) -- so the padding never reaches a weight gradient (0 x finite) or a bias gradient;
# * every op that does not know the layout goes through _f32c (one repack kernel) and hands back contiguous tensors.
# The model switches it on (channel_padded) only for geometries both plane kernels serve; HNO_PAD_ACT=0 switches it off (A/B).
_PAD = threading.local()


@contextlib.contextmanager
def channel_padded(on=True):
    prev = getattr(_PAD, 'on', False)
    _PAD.on = bool(on) and os.environ.get('HNO_PAD_ACT', '1') != '0'
    try:
        yield
    finally:
        _PAD.on = prev


def padded_ok(spatial, modes):
    """both transform directions take a padded channel stride for this grid / these (clamped) modes"""
    if len(spatial) != 3 or spatial[0] < 2 or int(np.prod(spatial)) % 32 == 0:
        return False
    m = clamp_modes(modes, spatial)
    return bool(_lib.lib().hno_dht3_ld_supported(*(int(v) for v in spatial), *(int(v) for v in m)))


def _pad_ld(V):
    return (int(V) + 31) // 32 * 32


def chan_stride(t):
    """channel stride (floats) when `t` is a channel-padded activation, else None"""
    if t is None or t.dim() != 5 or t.dtype != torch.float32 or t.is_meta:
        return None
    B, C, D, H, W = t.shape
    V, st = D * H * W, t.stride()
    ld = st[1]
    if ld == V or D < 2 or ld != _pad_ld(V) or tuple(st[2:]) != (H * W, W, 1) or (B > 1 and st[0] != C * ld):
        return None
    if (t.storage_offset() + B * C * ld) * 4 > t.untyped_storage().nbytes():
        return None
    return ld


def act_empty(B, C, spatial, device, ld=None):
    """uninitialised (B, C, *spatial) fp32 activation with channel stride `ld` (None: contiguous)"""
    D, H, W = (int(v) for v in spatial)
    if not ld or ld == D * H * W:
        return torch.empty((B, C, D, H, W), device=device, dtype=torch.float32)
    return torch.empty(B * C * ld, device=device, dtype=torch.float32).as_strided((B, C, D, H, W), (C * ld, ld, H * W, W, 1))


def act_like(t, C=None):
    return act_empty(t.shape[0], t.shape[1] if C is None else C, t.shape[2:], t.device, chan_stride(t))


# ---- bf16 activations in memory (round 6).  Under torch.autocast(bfloat16) the reference's nn.Conv3d returns bf16 and everything
# elementwise behind it stays bf16 (experiments/train_test.py:154-160): the input and output of every FNOSeg / HNOSeg block are bf16
# tensors there.  Here they are too -- same channel-padded layout, same ELEMENT stride ld (rows of 2-byte elements, 64-byte aligned) --
# so every pass over them moves half the bytes.  The operator output, the pre-concat activation y and all parameter gradients stay fp32
# as in the reference.  HNO_IO16=0 keeps fp32 storage (A/B).
def chan_stride16(t):
    """channel stride (elements) when `t` is a channel-padded bf16 activation, else None"""
    if t is None or t.dim() != 5 or t.dtype != torch.bfloat16 or t.is_meta:
        return None
    B, C, D, H, W = t.shape
    V, st = D * H * W, t.stride()
    ld = st[1]
    if D < 2 or ld != _pad_ld(V) or tuple(st[2:]) != (H * W, W, 1) or (B > 1 and st[0] != C * ld):
        return None
    if (t.storage_offset() + B * C * ld) * 2 > t.untyped_storage().nbytes() or (t.data_ptr() & 63):
        return None
    return ld


def act_empty16(B, C, spatial, device, ld):
    """uninitialised (B, C, *spatial) bf16 activation with channel stride `ld` elements"""
    D, H, W = (int(v) for v in spatial)
    return torch.empty(B * C * ld, device=device, dtype=torch.bfloat16).as_strided((B, C, D, H, W), (C * ld, ld, H * W, W, 1))


def to_bf16_layout(t, ld):
    """`t` (fp32 or bf16, any layout) as a channel-padded bf16 activation of stride `ld`; padding zero"""
    if chan_stride16(t) == ld:
        return t
    src = to_layout(t.float() if t.dtype != torch.float32 else t, ld)
    out = act_empty16(t.shape[0], t.shape[1], t.shape[2:], t.device, ld)
    check(_lib.lib().hno_cast_f32_bf16(ptr(src), ptr(out), _ext(src), stream_ptr()), 'hno_cast_f32_bf16')
    return out


def io16_enabled():
    return os.environ.get('HNO_IO16', '1') != '0'


def _ext(t):
    """floats an elementwise / pointwise kernel covers: the whole storage extent, padding included"""
    ld = chan_stride(t)
    return t.numel() if ld is None else t.shape[0] * t.shape[1] * ld


def to_layout(t, ld):
    """`t` with channel stride `ld` (None: contiguous); padding zeroed.  No-op when it already is."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    cur = chan_stride(t)
    if cur is None and not t.is_contiguous():
        t = t.contiguous()
    if cur == ld:
        return t
    B, C = t.shape[:2]
    V = _flat_v(t)
    out = act_empty(B, C, t.shape[2:], t.device, ld)
    check(_lib.lib().hno_chan_restride(ptr(t), ptr(out), B * C, V, cur or V, ld or V, stream_ptr()), 'hno_chan_restride')
    return out


def _f32a(t):
    """fp32 activation in a layout the pointwise kernels take: channel-padded as it is, anything else contiguous"""
    if t is None or (t.dtype == torch.float32 and chan_stride(t) is not None):
        return t
    return _f32c(t)


def clamp_modes(modes, spatial):
    """2m > s -> m = s // 2 (nets/hnosegxs.py:382-387)."""
    return tuple(int(s // 2 if 2 * m > s else m) for m, s in zip(modes, spatial))


# ------------------------------------------------------------------------- gradient destinations
# parallel.FlatGradReplica registers, per parameter, the slice of its flat all-reduce buffer that should receive the gradient:
# the backward kernels then write the weight gradient straight into the collective's buffer (no pack copy), and autograd
# installs that view as ``.grad``.  Keyed by the parameter's data pointer (saved tensors unpack to the same storage).
_grad_dest = {}
_dest_written = set()     # destinations handed out since the last new_grad_pass(): a second gradient of the same parameter
                          # (module applied twice, tied weights) must not overwrite the first -- it gets a fresh buffer


def set_grad_destinations(mapping):
    """mapping: {parameter: fp32 view of the same shape}; None / {} clears.  Returns the previous registry."""
    global _grad_dest
    old = _grad_dest
    _grad_dest = {p.data_ptr(): v for p, v in (mapping or {}).items()}
    _dest_written.clear()
    return old


def new_grad_pass():
    """call once per step before backward (FlatGradReplica.zero_grad does): every destination may be written once again"""
    _dest_written.clear()


def _grad_buffer(param, like=None):
    """where the gradient of `param` is to be written: its registered destination when one exists and nothing has been
    accumulated yet (``.grad`` is None or already that view, in which case the caller must NOT be accumulating), else fresh."""
    like = param if like is None else like
    if _grad_dest and param is not None:
        key = param.data_ptr()
        v = _grad_dest.get(key)
        if v is not None and tuple(v.shape) == tuple(like.shape) and param.grad is None and key not in _dest_written:
            _dest_written.add(key)
            # a FRESH tensor object on the destination's storage: AccumulateGrad installs an incoming gradient as .grad without a
            # copy only when nobody else references that tensor object -- handed the registered view itself (referenced by the
            # registry and the replica) it cloned every gradient, and the replica then copied the clone back (two copy kernels per
            # parameter and step, 0.3 ms of the HNOSeg-XS data-parallel step: profiles/r03_*)
            return v.detach()
    return torch.empty_like(like)


def _grad_buffer_stacked(params):
    """one (L, ...) buffer whose slices are the destinations of L same-shaped parameters, when those destinations lie back
    to back in that order in the flat buffer (they do: consecutive modules); else a fresh stacked tensor."""
    p0 = params[0]
    if _grad_dest:
        vs = [_grad_dest.get(p.data_ptr()) for p in params]
        if all(v is not None and p.grad is None and tuple(v.shape) == tuple(p.shape) and p.data_ptr() not in _dest_written
               for v, p in zip(vs, params)):
            n = p0.numel() * 4
            if all(vs[i].data_ptr() == vs[0].data_ptr() + i * n for i in range(len(vs))):
                _dest_written.update(p.data_ptr() for p in params)
                base = vs[0]._base if vs[0]._base is not None else vs[0]
                off = (vs[0].data_ptr() - base.data_ptr()) // 4
                return base.reshape(-1)[off:off + len(vs) * p0.numel()].view((len(vs),) + tuple(p0.shape))
    return torch.empty((len(params),) + tuple(p0.shape), device=p0.device, dtype=torch.float32)


def _grad_buffer_concat(params):
    """one flat fp32 buffer for the gradients of `params` laid end to end in this order: the registered destinations themselves when
    they lie back to back like that in the flat gradient buffer (consecutive parameters of consecutive modules do), else fresh."""
    n = sum(p.numel() for p in params)
    if _grad_dest:
        vs = [_grad_dest.get(p.data_ptr()) for p in params]
        if all(v is not None and p.grad is None and tuple(v.shape) == tuple(p.shape) and p.data_ptr() not in _dest_written
               for v, p in zip(vs, params)):
            off, ok = vs[0].data_ptr(), True
            for v in vs:
                ok = ok and v.data_ptr() == off
                off += v.numel() * 4
            if ok:
                _dest_written.update(p.data_ptr() for p in params)
                base = vs[0]._base if vs[0]._base is not None else vs[0]
                o = (vs[0].data_ptr() - base.data_ptr()) // 4
                return base.reshape(-1)[o:o + n]
    return torch.empty(n, device=params[0].device, dtype=torch.float32)


# ------------------------------------------------------------------------- raw launchers
def _block0(N0, m0):
    """size of the kept block along the first axis: a degenerate axis (N0 = 1, m0 = 0; 2-D data viewed as
    (B, C, 1, H, W)) keeps its single frequency."""
    return 1 if (N0 == 1 and m0 == 0) else 2 * m0


def dht3_crop_raw(x, modes, scale, act_out=None, act=ACT_NONE):
    """x: (B,C,N0,N1,N2) -> (B,C,2m0,2m1,2m2); modes already clamped."""
    _need_gpu(x, act_out)
    B, C, N0, N1, N2 = x.shape
    m0, m1, m2 = modes
    L = _lib.lib()
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, m0, m1, m2) // 4, device=x.device, dtype=torch.float32)
    out = torch.empty((B, C, _block0(N0, m0), 2 * m1, 2 * m2), device=x.device, dtype=torch.float32)
    ld = chan_stride(x)
    if ld is not None:
        act_out = to_layout(act_out, ld)
        check(L.hno_dht3_crop_ld(ptr(x), ptr(act_out), act if act_out is not None else ACT_NONE, ptr(out), ptr(ws),
                                 B * C, N0, N1, N2, m0, m1, m2, float(scale), ld, stream_ptr()), 'hno_dht3_crop_ld')
        return out
    check(L.hno_dht3_crop(ptr(x), ptr(act_out), act if act_out is not None else ACT_NONE, ptr(out), ptr(ws),
                          B * C, N0, N1, N2, m0, m1, m2, float(scale), stream_ptr()), 'hno_dht3_crop')
    return out


def dht3_full_raw(x, scale):
    """x: (B,C,N0,N1,N2) -> every frequency of the 3-D Hartley transform in natural order, any sizes
    (N0 = 1 gives the 2-D transform of each (N1, N2) plane)."""
    _need_gpu(x)
    B, C, N0, N1, N2 = x.shape
    L = _lib.lib()
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, N0 // 2, N1 // 2, N2 // 2) // 4, device=x.device,
                     dtype=torch.float32)
    out = torch.empty_like(x)
    check(L.hno_dht3_full(ptr(x), ptr(out), ptr(ws), B * C, N0, N1, N2, float(scale), stream_ptr()), 'hno_dht3_full')
    return out


def pad_idht3_raw(z, spatial, scale, addend=None, act=ACT_NONE, ld=None):
    """z: (B,C,2m0,2m1,2m2) -> (B,C,N0,N1,N2) = act(scale * IDHT(pad(z)) + addend); ld: channel stride of the result (and of the
    addend) when it is to be a channel-padded activation."""
    _need_gpu(z, addend)
    B, C = z.shape[:2]
    m0, m1, m2 = (s // 2 for s in z.shape[2:])
    N0, N1, N2 = (int(s) for s in spatial)
    L = _lib.lib()
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, m0, m1, m2) // 4, device=z.device, dtype=torch.float32)
    if ld is not None:
        addend = to_layout(addend, ld)
        out = act_empty(B, C, (N0, N1, N2), z.device, ld)
        check(L.hno_pad_idht3_ld(ptr(z), ptr(addend), act, ptr(out), ptr(ws), B * C, N0, N1, N2, m0, m1, m2,
                                 float(scale), ld, stream_ptr()), 'hno_pad_idht3_ld')
        return out
    out = torch.empty((B, C, N0, N1, N2), device=z.device, dtype=torch.float32)
    check(L.hno_pad_idht3(ptr(z), ptr(addend), act, ptr(out), ptr(ws), B * C, N0, N1, N2, m0, m1, m2,
                          float(scale), stream_ptr()), 'hno_pad_idht3')
    return out


def rfft3_crop_raw(x, modes, scale, k2_weights=False, act_out=None, act=ACT_NONE):
    """x (B,C,N0,N1,N2) -> kept half spectrum as real data (B, 2C, 2m0, 2m1, m2): channels [re | im]."""
    _need_gpu(x, act_out)
    B, C, N0, N1, N2 = x.shape
    m0, m1, m2 = modes
    L = _lib.lib()
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, m0, m1, m2) // 4, device=x.device, dtype=torch.float32)
    out = torch.empty((B, 2 * C, _block0(N0, m0), 2 * m1, m2), device=x.device, dtype=torch.float32)
    ld = chan_stride(x)
    if ld is not None:
        act_out = to_layout(act_out, ld)
        check(L.hno_rfft3_crop_ld(ptr(x), ptr(act_out), act if act_out is not None else ACT_NONE, ptr(out), ptr(ws), B, C,
                                  N0, N1, N2, m0, m1, m2, float(scale), int(k2_weights), ld, stream_ptr()), 'hno_rfft3_crop_ld')
        return out
    check(L.hno_rfft3_crop(ptr(x), ptr(act_out), act if act_out is not None else ACT_NONE, ptr(out), ptr(ws), B, C,
                           N0, N1, N2, m0, m1, m2, float(scale), int(k2_weights), stream_ptr()), 'hno_rfft3_crop')
    return out


def irfft3_pad_raw(spec, spatial, scale, k2_weights=True, addend=None, act=ACT_NONE, ld=None):
    """spec (B, 2C, 2m0, 2m1, m2) -> (B, C, N0, N1, N2) = act(scale * irfft-style inverse + addend)."""
    _need_gpu(spec, addend)
    B, C2 = spec.shape[:2]
    C = C2 // 2
    m0, m1, m2 = spec.shape[2] // 2, spec.shape[3] // 2, spec.shape[4]
    N0, N1, N2 = (int(v) for v in spatial)
    L = _lib.lib()
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, m0, m1, m2) // 4, device=spec.device, dtype=torch.float32)
    if ld is not None:
        addend = to_layout(addend, ld)
        out = act_empty(B, C, (N0, N1, N2), spec.device, ld)
        check(L.hno_irfft3_pad_ld(ptr(spec), ptr(addend), act, ptr(out), ptr(ws), B, C, N0, N1, N2, m0, m1, m2, float(scale),
                                  int(k2_weights), ld, stream_ptr()), 'hno_irfft3_pad_ld')
        return out
    out = torch.empty((B, C, N0, N1, N2), device=spec.device, dtype=torch.float32)
    check(L.hno_irfft3_pad(ptr(spec), ptr(addend), act, ptr(out), ptr(ws), B, C, N0, N1, N2, m0, m1, m2, float(scale),
                           int(k2_weights), stream_ptr()), 'hno_irfft3_pad')
    return out


def act_bwd_raw(g, y, act):
    gx = torch.empty_like(g)
    check(_lib.lib().hno_act_bwd(ptr(g), ptr(y), ptr(gx), g.numel(), act, stream_ptr()), 'hno_act_bwd')
    return gx


def _flat_v(t):
    return int(np.prod(t.shape[2:]))


def _wgrad_ws(cin, cout, device):
    n = _lib.lib().hno_pwconv_bwd_workspace_bytes(int(cin), int(cout)) // 4
    return torch.empty(n, device=device, dtype=torch.float32)


def _autocast_bf16():
    """True inside torch.autocast('cuda', dtype=bfloat16): the pointwise kernels then run their channel contraction on the bf16
    matrix cores (24 / 48-channel shapes; see ops_bf16.autocast_bf16)"""
    if not torch.is_autocast_enabled('cuda'):
        return False
    from . import ops_bf16
    return ops_bf16.autocast_bf16()


def pwconv_fwd_raw(xa, xb, W, bias, act, bf16=False):
    B, Ca = xa.shape[:2]
    Cb = xb.shape[1] if xb is not None else 0
    ld = chan_stride(xa)            # channel-padded operands: the kernel runs over V := ld voxels per channel
    xb = to_layout(xb, ld)
    Cout, V = W.shape[0], ld or _flat_v(xa)
    assert W.numel() == Cout * (Ca + Cb), 'weight shape does not match the concatenated input channels'
    y = act_like(xa, Cout)
    check(_lib.lib().hno_pwconv_fwd(ptr(xa), Ca, ptr(xb), Cb, ptr(W), ptr(bias), ptr(y), B, Cout, V, act | (ACT_BF16 if bf16 else 0), stream_ptr()),
          'hno_pwconv_fwd')
    return y


def pwconv_bwd_raw(gy, y, xa, xb, W, act, has_bias, need_gxa=True, need_gxb=True, xa_act=ACT_NONE, accumulate_into=None,
                   defer=False, bias=None, bf16=False):
    """-> (gxa, gxb, dW, dbias); y is the saved output (None when act is NONE).  xa_act: also multiply
    gxa by act'(xa) (xa being the output of that activation)."""
    B, Ca = xa.shape[:2]
    Cb = xb.shape[1] if xb is not None else 0
    ld = chan_stride(xa)            # channel-padded operands (all of them): V := ld; gy's padding is zero, so is gxa's / gxb's
    gy, y, xb = to_layout(gy, ld), to_layout(y, ld), to_layout(xb, ld)
    Cout, V = W.shape[0], ld or _flat_v(xa)
    acc_bits = 0
    gxa = gxb = None
    if accumulate_into is not None:      # (gxa, gxb) buffers that already hold a gradient (None: fresh): += fused into the store
        gxa, gxb = accumulate_into
        assert all(t is None or chan_stride(t) == ld for t in (gxa, gxb)), 'accumulation buffer in another layout'
        acc_bits = (1 if gxa is not None else 0) | (2 if gxb is not None else 0)
    if gxa is None and need_gxa:
        gxa = act_like(xa)
    if gxb is None and xb is not None and need_gxb:
        gxb = act_like(xb)
    dW = _grad_buffer(W)
    db = None
    if has_bias:
        db = _grad_buffer(bias) if bias is not None else torch.empty(Cout, device=W.device, dtype=torch.float32)
    ws = _wgrad_ws(Ca + Cb, Cout, xa.device)
    with _DeferReduce(defer) as d:
        check(_lib.lib().hno_pwconv_bwd(ptr(gy), ptr(y), ptr(xa), Ca, ptr(xb), Cb, ptr(W), ptr(gxa), ptr(gxb), ptr(dW),
                                        ptr(db), ptr(ws), B, Cout, V, act | (ACT_BF16 if bf16 else 0), xa_act, acc_bits | d.bit, stream_ptr()), 'hno_pwconv_bwd')
        d.keep(ws)
    return gxa, gxb, dW, db


def pwconv_bwd_branch_raw(gy, y, xa, xb, W, Wbr, act, xa_act, defer=False, bf16=False, io16=False):
    """Backward of  act(W [xa ; xb] + b)  where xa = xa_act(s + Wbr xb + bbr): one pass (hno_pwconv_bwd_branch).
    -> (p, gxb, dW, db, dWbr, dbbr) with p the gradient of the pre-activation sum s + Wbr xb + bbr.
    io16: gy, y, xb are channel-padded bf16 tensors (xa's stride); p and gxb come back fp32."""
    B, Ca = xa.shape[:2]
    ld = chan_stride(xa)            # channel-padded operands: V := ld (pwconv_bwd_raw)
    if io16:
        assert bf16 and ld is not None
        gy, y, xb = to_bf16_layout(gy, ld), to_bf16_layout(y, ld), to_bf16_layout(xb, ld)
    else:
        gy, y, xb = to_layout(gy, ld), to_layout(y, ld), to_layout(xb, ld)
    Cb, Cout, V = xb.shape[1], W.shape[0], ld or _flat_v(xa)
    L = _lib.lib()
    p = act_like(xa)
    gxb = act_like(xa, Cb)
    n_w, n_br = Cout * (Ca + Cb), Ca * Cb
    flat = torch.empty(n_w + Cout + n_br + Ca, device=xa.device, dtype=torch.float32)
    ws = torch.empty(L.hno_pwconv_bwd_branch_workspace_bytes(Ca, Cb, Cout) // 4, device=xa.device, dtype=torch.float32)
    with _DeferReduce(defer) as d:
        check(L.hno_pwconv_bwd_branch(ptr(gy), ptr(y), ptr(xa), Ca, ptr(xb), Cb, ptr(W), ptr(Wbr), ptr(p), ptr(gxb), ptr(flat),
                                      ptr(ws), B, Cout, V, act | (ACT_BF16 if bf16 else 0) | (ACT_IO16 if io16 else 0), xa_act | d.bit, stream_ptr()),
              'hno_pwconv_bwd_branch')
        d.keep(ws)
    dW = flat[:n_w].view_as(W)
    db = flat[n_w:n_w + Cout]
    dWbr = flat[n_w + Cout:n_w + Cout + n_br].view(Ca, Cb)
    dbbr = flat[n_w + Cout + n_br:]
    return p, gxb, dW, db, dWbr, dbbr


# ---- deferred slab reductions (include/hno.h: hno_set_defer_reduce) -------------------------------------------------
# During autograd's backward the weight-gradient reductions of all layers can be recorded and launched as ONE kernel from
# the engine's end-of-backward callback; the slab workspaces are kept alive here until then.  The dW tensor handed to
# autograd is uninitialised until that callback runs, so this is OPT-IN (set_defer_reduce(True) or HNO_DEFER_REDUCE=1;
# bench.py enables it; training() leaves the setting alone, and FlatGradReplica switches it off while its per-bucket hooks send
# gradients during backward) and guarded: a gradient is only deferred when nothing can read it before the
# pass ends -- the weight is a leaf with .grad None, has no tensor / post-accumulate hooks, and feeds exactly ONE live
# autograd node (a module applied twice, or tied weights, would make AccumulateGrad sum two unreduced tensors).
import os as _os
import weakref as _weakref
_DEFER_ENABLED = bool(int(_os.environ.get('HNO_DEFER_REDUCE', '0')))
_defer_state = {'active': False, 'keep': [], 'task': None, 'stream': None}
_stats = {'pass_fused': 0, 'pass_unfused': 0}     # counters read by the tests
_param_uses = {}     # id(parameter) -> [weak references to the live autograd nodes (ctx) holding it, poisoned]


def set_defer_reduce(on=True):
    """Enable / disable the batched end-of-backward weight-gradient reduction (see above).  Returns the old setting."""
    global _DEFER_ENABLED
    old, _DEFER_ENABLED = _DEFER_ENABLED, bool(on)
    return old


def _flush_deferred():
    L = _lib.lib()
    L.hno_set_defer_reduce(0)
    try:
        if L.hno_pending_reduces():
            # on the stream the slab writers ran on (the engine may run this callback on another one)
            st = _defer_state['stream']
            if st is not None and st != torch.cuda.current_stream():
                with torch.cuda.stream(st):
                    check(L.hno_flush_reduces(stream_ptr()), 'hno_flush_reduces')
                torch.cuda.current_stream().wait_stream(st)
            else:
                check(L.hno_flush_reduces(stream_ptr()), 'hno_flush_reduces')
    finally:
        _defer_reset()


def _defer_reset():
    _defer_state['active'] = False
    _defer_state['task'] = None
    _defer_state['stream'] = None
    _defer_state['keep'].clear()


def _defer_drop_stale():
    """Called outside any backward pass: records still pending belong to a pass that raised before its callback ran
    (their workspaces are kept alive in 'keep', ~100 MB): drop both."""
    if _defer_state['active'] and torch._C._current_graph_task_id() < 0:
        _lib.lib().hno_discard_reduces()
        _lib.lib().hno_set_defer_reduce(0)
        _defer_reset()


def _note_use(ctx, *weights):
    """forward-time: these parameters are held by one more live autograd node.  The node is remembered through a weak reference to
    its ctx: a forward that never gets a backward (torch.no_grad() -- ctx.needs_input_grad is True there too --, validation passes,
    an exception before backward) leaves no trace once its outputs are gone.  (Round 4 counted uses instead: one no_grad forward of a
    model -- the half-batch warm-up in front of a capture, any validation epoch -- left every parameter "held twice" for good, and
    the model's passes reduced their slabs one launch at a time from then on.)"""
    if not _DEFER_ENABLED:
        return
    ref = _weakref.ref(ctx)
    for w in weights:
        if w is not None and w.requires_grad:
            e = _param_uses.setdefault(id(w), [[], False])
            e[0] = [r for r in e[0] if r() is not None]
            if not e[0]:
                e[1] = False
            e[0].append(ref)
            if len(e[0]) > 1:
                e[1] = True


def _release_use(ctx, *weights):
    """backward-time: -> True when every one of these parameters fed only this node since it was last idle"""
    ok = True
    for w in weights:
        if w is None:
            continue
        e = _param_uses.get(id(w))
        if e is None:
            ok = False
            continue
        live = [r for r in e[0] if r() is not None and r() is not ctx]
        if len(live) + 1 != len(e[0]):          # dead nodes dropped: were they what poisoned the entry?
            e[1] = e[1] and len(live) > 0
        ok = ok and not e[1] and not live
        e[0] = live
        if not live:
            del _param_uses[id(w)]
    return ok


def _deferrable(*weights):
    """A weight gradient may be produced late only if nothing reads it before the end of backward: the weight is a leaf
    whose .grad is None (autograd then just installs the tensor), not a slice / cat / reshape of parameters (whose backward
    nodes read the gradient), has no hooks that would see the unreduced tensor, and is not accumulated with the gradient of
    a second use (checked by the callers through _release_use)."""
    for w in weights:
        if w is None:
            continue
        if not (w.is_leaf and w.grad is None) or w._backward_hooks or getattr(w, '_post_accumulate_grad_hooks', None):
            return False
    return True


def _leaf_params(ctx, *ts):
    """forward-time half of the check: the tensors handed in ARE leaves, used as they are (no fp32 / contiguous copy).
    When the node will run a backward (some input needs a gradient; Function.forward itself runs with grad mode off, so
    ctx.needs_input_grad is the signal) the parameters are counted as held by one more live node (_note_use)."""
    ok = all(t is None or (t.is_leaf and t.dtype == torch.float32 and t.is_contiguous()) for t in ts)
    if ok and ctx is not None and any(ctx.needs_input_grad):
        _note_use(ctx, *ts)
        return True
    return False


class _DeferReduce:
    """with _DeferReduce(ok) as d: <raw backward call>; d.keep(ws).  Defers only inside an autograd backward pass and only
    when the caller vouches (``ok``) that the gradient is not read before the pass ends."""

    def __init__(self, ok=False):
        self.ok = ok

    def __enter__(self):
        self.on = False
        if _DEFER_ENABLED and self.ok:
            task = torch._C._current_graph_task_id()      # -1 outside backward(); one id per backward pass
            if task < 0:
                return self                                # not inside backward(): reduce immediately
            if _defer_state['active'] and _defer_state['task'] != task:
                return self                                # a nested (re-entrant) pass: leave the outer pass's records alone
            if not _defer_state['active']:
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred)
                    _defer_state['active'], _defer_state['task'] = True, task
                    _defer_state['stream'] = torch.cuda.current_stream()
                except RuntimeError:
                    return self
            self.on = True
        return self

    @property
    def bit(self):
        """0x100 when this call's reduction is to be recorded (ORed into the int argument that carries it)"""
        return 0x100 if self.on else 0

    def keep(self, *tensors):
        if self.on:
            _defer_state['keep'].extend(tensors)

    def __exit__(self, *exc):
        return False


def _layer_ptrs(Ws):
    import ctypes
    return (ctypes.c_void_p * len(Ws))(*[w.data_ptr() for w in Ws])


def _mix_layers(W):
    """(L, C, C) tensor or a sequence of L (C, C) tensors -> list of contiguous fp32 (C, C) tensors."""
    if torch.is_tensor(W):
        W = W.unbind(0)
    return [_f32c(w) for w in W]


def specmix_fwd_raw(z0, W, residual, act):
    Ws = _mix_layers(W)
    B, C = z0.shape[:2]
    M, Lyr = _flat_v(z0), len(Ws)
    zs = torch.empty((Lyr,) + tuple(z0.shape), device=z0.device, dtype=torch.float32)
    check(_lib.lib().hno_specmix_layers_fwd(ptr(z0), _layer_ptrs(Ws), ptr(zs), B, C, M, Lyr, int(residual), act, stream_ptr()),
          'hno_specmix_layers_fwd')
    return zs


def spectral_chain_supported(x, modes, n_layers):
    """fused spectral middle (hno_spec_mid_*): HNOSeg-XS shapes on 65^3 / 33^3 grids; HNO_FUSED_MID=0 switches it off (A/B)."""
    if os.environ.get('HNO_FUSED_MID', '1') == '0' or x.dim() != 5:
        return False
    m0, m1, m2 = modes
    return bool(_lib.lib().hno_spec_mid_supported(int(x.shape[1]), int(x.shape[2]), int(m0), int(m1), int(m2), int(n_layers)))


def spectral_chain_fwd_raw(x, W, modes, act, scale_fwd, inv_act, residual=1, addend=None):
    """TransformCrop -> n_XS frequency-domain layers z <- act((W [+ I]) z) -> PadInverse (+ addend, + activation) with the fused
    middle: -> (z0, zs, u): what dht3_crop_raw / specmix_fwd_raw / pad_idht3_raw return."""
    _need_gpu(x, addend)
    Ws = _mix_layers(W)
    B, C, N0, N1, N2 = x.shape
    m0, m1, m2 = modes
    L, Lyr = _lib.lib(), len(Ws)
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, m0, m1, m2) // 4, device=x.device, dtype=torch.float32)
    zall = torch.empty((Lyr + 1, B, C, 2 * m0, 2 * m1, 2 * m2), device=x.device, dtype=torch.float32)
    ld = chan_stride(x) or 0
    u = act_like(x)
    check(L.hno_dht3_planes(ptr(x), ptr(ws), B * C, N0, N1, N2, m0, m1, m2, ld, stream_ptr()), 'hno_dht3_planes')
    addend = to_layout(addend, ld or None)
    check(L.hno_spec_mid_fwd(ptr(ws), _layer_ptrs(Ws), ptr(zall), B, C, N0, m0, m1, m2, Lyr, int(residual), act, float(scale_fwd), stream_ptr()),
          'hno_spec_mid_fwd')
    check(L.hno_idht3_planes(ptr(ws), ptr(addend), inv_act, ptr(u), B * C, N0, N1, N2, m0, m1, m2, 1.0, ld, stream_ptr()), 'hno_idht3_planes')
    return zall[0], zall[1:], u


def spectral_chain_bwd_ok(xm, modes, z0, zs):
    """backward of the fused spectral middle: same configurations as the forward, and z_0 .. z_L stacked in one buffer (what
    spectral_chain_fwd_raw returns); HNO_FUSED_MID_BWD=0 switches it off (A/B)."""
    return (os.environ.get('HNO_FUSED_MID_BWD', '1') != '0' and spectral_chain_supported(xm, modes, zs.shape[0])
            and zs.is_contiguous() and z0.is_contiguous() and zs.data_ptr() == z0.data_ptr() + 4 * z0.numel())


def spectral_chain_bwd_raw(g_u, z0, W, modes, act, scale_out, addend, defer=False, residual=1):
    """PadInverse^T -> backward of the n_XS frequency-domain layers -> TransformCrop^T (+ addend) with the fused middle: g_u is the
    gradient of PadInverse's pre-activation output, z0 the base of the stacked z_0 .. z_L of the forward; -> (g_xm, dW (L, C, C)).
    The gradients of the cropped spectra never reach memory."""
    _need_gpu(g_u, addend)
    Ws = _mix_layers(W)
    B, C, N0, N1, N2 = g_u.shape
    m0, m1, m2 = modes
    L, Lyr = _lib.lib(), len(Ws)
    ld = chan_stride(g_u)
    addend = to_layout(addend, ld)
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, m0, m1, m2) // 4, device=g_u.device, dtype=torch.float32)
    slab = torch.empty(L.hno_spec_mid_bwd_workspace_bytes(B, C, m1, Lyr) // 4, device=g_u.device, dtype=torch.float32)
    dW = _grad_buffer_stacked(W) if not torch.is_tensor(W) else torch.empty((Lyr, C, C), device=g_u.device, dtype=torch.float32)
    g_xm = act_like(g_u)
    check(L.hno_dht3_planes(ptr(g_u), ptr(ws), B * C, N0, N1, N2, m0, m1, m2, ld or 0, stream_ptr()), 'hno_dht3_planes')
    with _DeferReduce(defer) as d:
        check(L.hno_spec_mid_bwd(ptr(ws), _layer_ptrs(Ws), ptr(z0), ptr(dW), ptr(slab), 4 * slab.numel(), B, C, N0, m0, m1, m2, Lyr,
                                 int(residual) | d.bit, act, 1.0,
                                 stream_ptr()), 'hno_spec_mid_bwd')
        d.keep(slab)
    check(L.hno_idht3_planes(ptr(ws), ptr(addend), ACT_NONE, ptr(g_xm), B * C, N0, N1, N2, m0, m1, m2, float(scale_out), ld or 0,
                             stream_ptr()), 'hno_idht3_planes')
    return g_xm, dW


def cmix_compose_all(pairs):
    """[(wr, wi), ...] of equal shape (Co, Ci) -> (n, 2Co, 2Ci): the composed real forms W2 = [[wr, -wi], [wi, wr]] of all Fourier blocks
    of a model in ONE launch (hno_cmix_compose_multi); None when the pairs do not qualify (the blocks then compose their own)."""
    import ctypes
    if not pairs or len(pairs) > 64:
        return None
    shp = tuple(pairs[0][0].shape)
    for wr, wi in pairs:
        if not (wr.is_cuda and wi.is_cuda and wr.dtype == wi.dtype == torch.float32 and wr.is_contiguous() and wi.is_contiguous()
                and tuple(wr.shape) == tuple(wi.shape) == shp and len(shp) == 2):
            return None
    Co, Ci = shp
    out = torch.empty((len(pairs), 2 * Co, 2 * Ci), device=pairs[0][0].device, dtype=torch.float32)
    arr = ctypes.c_void_p * len(pairs)
    check(_lib.lib().hno_cmix_compose_multi(arr(*[w.data_ptr() for w, _ in pairs]), arr(*[w.data_ptr() for _, w in pairs]), ptr(out),
                                           len(pairs), Co, Ci, stream_ptr()), 'hno_cmix_compose_multi')
    return out


def fourier_chain_supported(x, modes):
    """fused middle of the Fourier block (hno_spec_mid_fourier_*): 24 channels on 65^3 / 33^3 grids; HNO_FUSED_MID=0 switches it off."""
    if os.environ.get('HNO_FUSED_MID', '1') == '0' or x.dim() != 5:
        return False
    m0, m1, m2 = modes
    return bool(_lib.lib().hno_spec_mid_fourier_supported(int(x.shape[1]), int(x.shape[2]), int(m0), int(m1), int(m2)))


def fourier_chain_fwd_raw(x, w2, modes, scale_fwd, addend, inv_act):
    """rfftn + mode selection -> complex channel mix (w2: the composed real (2C, 2C) form) -> zero pad + irfftn (+ addend, + activation)
    with the fused middle: -> (s0, y) = what rfft3_crop_raw returns and irfft3_pad_raw(pwconv(s0, w2), ...) returns.
    x may be a channel-padded bf16 activation (chan_stride16): its planes are read as they are (hno_dht3_planes_b16); y is fp32."""
    _need_gpu(x, addend)
    B, C, N0, N1, N2 = x.shape
    m0, m1, m2 = modes
    L = _lib.lib()
    x16 = chan_stride16(x)
    ld = x16 or chan_stride(x) or 0
    addend = to_layout(addend, ld or None)
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, m0, m1, m2) // 4, device=x.device, dtype=torch.float32)
    s0 = torch.empty((B, 2 * C, 2 * m0, 2 * m1, m2), device=x.device, dtype=torch.float32)
    y = act_empty(B, C, (N0, N1, N2), x.device, ld or None)
    if x16:
        check(L.hno_dht3_planes_b16(ptr(x), ptr(ws), B * C, N0, N1, N2, m0, m1, m2, ld, stream_ptr()), 'hno_dht3_planes_b16')
    else:
        check(L.hno_dht3_planes(ptr(x), ptr(ws), B * C, N0, N1, N2, m0, m1, m2, ld, stream_ptr()), 'hno_dht3_planes')
    check(L.hno_spec_mid_fourier_fwd(ptr(ws), ptr(w2), ptr(s0), B, C, N0, m0, m1, m2, float(scale_fwd), 0, 1, stream_ptr()),
          'hno_spec_mid_fourier_fwd')
    check(L.hno_idht3_planes(ptr(ws), ptr(addend), inv_act, ptr(y), B * C, N0, N1, N2, m0, m1, m2, 1.0, ld, stream_ptr()), 'hno_idht3_planes')
    return s0, y


def fourier_chain_bwd_raw(p, s0, w2, modes, scale_out, addend, defer=False, out16=False):
    """backward of the same chain: p = gradient of the inverse transform's (pre-activation) output -> (gx, dW2 (2C, 2C)); the gradients
    of the spectra never reach memory.  defer: dW2's slab reduction joins the batched end-of-backward reduction (the caller then
    defers the real / imaginary split likewise: cmix_split_grad_raw).  out16: gx is the gradient of a bf16 block input -- written as a
    channel-padded bf16 tensor by the inverse plane kernel's epilogue (hno_idht3_planes_b16; the addend stays fp32)."""
    _need_gpu(p, addend)
    B, C, N0, N1, N2 = p.shape
    m0, m1, m2 = modes
    L = _lib.lib()
    ld = chan_stride(p) or 0
    addend = to_layout(addend, ld or None)
    ws = torch.empty(L.hno_dht3_workspace_bytes(B * C, N0, N1, N2, m0, m1, m2) // 4, device=p.device, dtype=torch.float32)
    slab = torch.empty(L.hno_spec_mid_fourier_bwd_workspace_bytes(B, C, m1) // 4, device=p.device, dtype=torch.float32)
    dw2 = torch.empty((2 * C, 2 * C), device=p.device, dtype=torch.float32)
    gx = act_empty16(B, C, (N0, N1, N2), p.device, ld) if (out16 and ld) else act_like(p)
    check(L.hno_dht3_planes(ptr(p), ptr(ws), B * C, N0, N1, N2, m0, m1, m2, ld, stream_ptr()), 'hno_dht3_planes')
    with _DeferReduce(defer) as d:
        check(L.hno_spec_mid_fourier_bwd(ptr(ws), ptr(w2), ptr(s0), ptr(dw2), ptr(slab), 4 * slab.numel(), B, C, N0, m0, m1, m2, 1.0, 1 | d.bit, 0,
                                         stream_ptr()),
              'hno_spec_mid_fourier_bwd')
        d.keep(slab, dw2)
    if gx.dtype == torch.bfloat16:
        check(L.hno_idht3_planes_b16(ptr(ws), ptr(addend), ACT_NONE, ptr(gx), B * C, N0, N1, N2, m0, m1, m2, float(scale_out), ld, stream_ptr()),
              'hno_idht3_planes_b16')
    else:
        check(L.hno_idht3_planes(ptr(ws), ptr(addend), ACT_NONE, ptr(gx), B * C, N0, N1, N2, m0, m1, m2, float(scale_out), ld, stream_ptr()),
              'hno_idht3_planes')
    return gx, dw2


def specmix_bwd_raw(g, z0, zs, W, residual, act, defer=False):
    Ws = _mix_layers(W)
    B, C = z0.shape[:2]
    M, Lyr = _flat_v(z0), len(Ws)
    if residual and C > 32:
        # the kernels' residual (W + I) form is built for <= 32 channels; wider layers (the reference takes any `filters`) get W + I as an
        # explicit matrix: d loss / d (W + I) = d loss / d W, everything else is the plain form (round 6: this used to raise)
        eye = torch.eye(C, device=z0.device, dtype=torch.float32)
        Ws, residual = [w + eye for w in Ws], 0
    gz0 = torch.empty_like(z0)
    dW = _grad_buffer_stacked(W) if not torch.is_tensor(W) else torch.empty((Lyr, C, C), device=z0.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().hno_specmix_bwd_workspace_bytes(B, C, M, Lyr) // 4, device=z0.device, dtype=torch.float32)
    with _DeferReduce(defer) as d:
        check(_lib.lib().hno_specmix_layers_bwd(ptr(g), ptr(z0), ptr(zs), _layer_ptrs(Ws), ptr(gz0), ptr(dW), ptr(ws), B, C, M, Lyr,
                                                int(residual) | d.bit, act, stream_ptr()), 'hno_specmix_layers_bwd')
        d.keep(ws)
    return gz0, dW


# ----------------------------------------------------------------------------- autograd
_outer_grad = {'enabled': True}


def backward_wanted(ctx):
    """a backward pass can follow this forward: grad mode was on at the call AND some input requires grad"""
    return _outer_grad['enabled'] and any(ctx.needs_input_grad)


class _HnoFunction(torch.autograd.Function):
    """Base of every fused op.  Tensors on the ``meta`` device carry shapes only: ``apply`` then returns empty meta
    tensors of the output shapes (``cls.meta``) without touching libhno -- this is what the reference's
    ``save_model_summary(copy.deepcopy(model), input_size)`` / ``torchview.draw_graph(..., device='meta')`` need
    (experiments/utils.py:122-134, train_test.py:117-122).  It is shape inference, not a compute path: any real CPU
    tensor still raises (``_need_gpu``)."""

    @classmethod
    def apply(cls, *args):
        if any(torch.is_tensor(a) and a.is_meta for a in args):
            return cls.meta(*args)
        if _defer_state['active']:
            _defer_drop_stale()
        # torch.is_grad_enabled() is always False inside Function.forward and ctx.needs_input_grad reflects requires_grad, not the
        # grad mode: the caller's grad mode is recorded here for forwards that skip work only a backward would read
        prev = _outer_grad['enabled']
        _outer_grad['enabled'] = torch.is_grad_enabled()
        try:
            return super().apply(*args)
        finally:
            _outer_grad['enabled'] = prev

    @staticmethod
    def meta(*args):
        raise NotImplementedError


def _m(shape, dtype=torch.float32):
    return torch.empty(tuple(int(v) for v in shape), device='meta', dtype=dtype)

class DhtFullFn(_HnoFunction):
    """Un-truncated dhtn (nets/dht.py:16-49).  The Hartley matrix is symmetric, so backward is the same transform."""

    @staticmethod
    def meta(x, scale):
        return _m(x.shape)

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return dht3_full_raw(_f32c(x), scale)

    @staticmethod
    def backward(ctx, g):
        return dht3_full_raw(_f32c(g), ctx.scale), None


class DhtCropFn(_HnoFunction):
    """TransformCrop (nets/hnosegxs.py:378-410).  backward = PadInverse * scale."""

    @staticmethod
    def meta(x, modes, scale):
        return _m(tuple(x.shape[:2]) + (_block0(x.shape[2], modes[0]), 2 * modes[1], 2 * modes[2]))

    @staticmethod
    def forward(ctx, x, modes, scale):
        x = _f32c(x)
        ctx.spatial, ctx.scale = tuple(x.shape[2:]), scale
        return dht3_crop_raw(x, modes, scale)

    @staticmethod
    def backward(ctx, g):
        return pad_idht3_raw(_f32c(g), ctx.spatial, ctx.scale), None, None


class PadIdhtFn(_HnoFunction):
    """PadInverse (nets/hnosegxs.py:454-494) with fused output activation.
    backward = TransformCrop of (g * act'(out)) with the same scale."""

    @staticmethod
    def meta(z, spatial, scale, act):
        return _m(tuple(z.shape[:2]) + tuple(spatial))

    @staticmethod
    def forward(ctx, z, spatial, scale, act):
        z = _f32c(z)
        out = pad_idht3_raw(z, spatial, scale, None, act)
        ctx.modes, ctx.scale, ctx.act = tuple(s // 2 for s in z.shape[2:]), scale, act
        if act != ACT_NONE:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        out = ctx.saved_tensors[0] if ctx.act != ACT_NONE else None
        return dht3_crop_raw(_f32c(g), ctx.modes, ctx.scale, out, ctx.act), None, None, None


_PADADD_BWD_ONE_PRODUCT = os.environ.get('HNO_PADADD_BWD_ONE_PRODUCT', '1') != '0'


class PadIdhtAddFn(_HnoFunction):
    """act(scale * PadInverse(z) + addend): the operator output plus the spatial conv branch, activated
    (nets/architectures.py:521-539 with HartleyOperator._call3d :243-269 feeding it)."""

    @staticmethod
    def meta(z, addend, spatial, scale, act):
        return _m(tuple(z.shape[:2]) + tuple(spatial))

    @staticmethod
    def forward(ctx, z, addend, spatial, scale, act):
        z, addend = _f32c(z), _f32c(addend)
        out = pad_idht3_raw(z, spatial, scale, addend, act)
        ctx.modes, ctx.scale, ctx.act = tuple(s // 2 for s in z.shape[2:]), scale, act
        ctx.save_for_backward(out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = _f32c(g)
        if ctx.act != ACT_NONE and ctx.needs_input_grad[1] and _PADADD_BWD_ONE_PRODUCT:
            # the addend's gradient g * act'(out) is needed as a tensor anyway: transform THAT instead of letting the transform form the
            # product again from g and out -- one input stream less, and the plain plane kernel (LDS-DMA rows) instead of the two-operand
            # one (HartleyMHASeg: 20.1 -> 8.9 us per block)
            ga = act_bwd_raw(g, out, ctx.act)
            return dht3_crop_raw(ga, ctx.modes, ctx.scale), ga, None, None, None
        gz = dht3_crop_raw(g, ctx.modes, ctx.scale, out, ctx.act)
        ga = None
        if ctx.needs_input_grad[1]:
            ga = act_bwd_raw(g, out, ctx.act) if ctx.act != ACT_NONE else g
        return gz, ga, None, None, None


class RfftCropFn(_HnoFunction):
    """rfftn(norm='forward') + corner gather (nets/fourier_operator.py:164-191) -> (B, 2C, 2m0, 2m1, m2)."""

    @staticmethod
    def meta(x, modes):
        return _m((x.shape[0], 2 * x.shape[1], _block0(x.shape[2], modes[0]), 2 * modes[1], modes[2]))

    @staticmethod
    def forward(ctx, x, modes):
        x = _f32c(x)
        ctx.spatial = tuple(x.shape[2:])
        ctx.scale = 1.0 / float(np.prod(ctx.spatial))
        return rfft3_crop_raw(x, modes, ctx.scale, False)

    @staticmethod
    def backward(ctx, g):
        return irfft3_pad_raw(_f32c(g), ctx.spatial, ctx.scale, False), None


class IrfftPadFn(_HnoFunction):
    """act(zero-pad + irfftn(norm='forward') + addend) (nets/fourier_operator.py:195-209)."""

    @staticmethod
    def meta(spec, addend, spatial, act):
        return _m((spec.shape[0], spec.shape[1] // 2) + tuple(spatial))

    @staticmethod
    def forward(ctx, spec, addend, spatial, act):
        spec, addend = _f32c(spec), _f32c(addend)
        out = irfft3_pad_raw(spec, spatial, 1.0, True, addend, act)
        ctx.modes, ctx.act = (spec.shape[2] // 2, spec.shape[3] // 2, spec.shape[4]), act
        ctx.save_for_backward(out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = _f32c(g)
        gs = rfft3_crop_raw(g, ctx.modes, 1.0, True, out, ctx.act)
        ga = None
        if ctx.needs_input_grad[1]:
            ga = act_bwd_raw(g, out, ctx.act) if ctx.act != ACT_NONE else g
        return gs, ga, None, None


def _conv_ws(cin, cout, wgrad, device):
    n = _lib.lib().hno_conv3d_k3_workspace_bytes(int(cin), int(cout), int(wgrad)) // 4
    return torch.empty(n, device=device, dtype=torch.float32)


def _chan_sum(g):
    B, C = g.shape[:2]
    out = torch.empty(C, device=g.device, dtype=torch.float32)
    L = _lib.lib()
    ws = torch.empty(L.hno_channel_sum_workspace_bytes(C) // 8, device=g.device, dtype=torch.float64)
    check(L.hno_channel_sum(ptr(g), ptr(out), ptr(ws), B, C, _flat_v(g), stream_ptr()), 'hno_channel_sum')
    return out


def _conv3d_call(x, W, bias, out_shape, mode, cin, cout, stride, act=ACT_NONE):
    y = torch.empty(out_shape, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    nbytes = L.hno_conv3d_k3_fwd_workspace_bytes(mode, x.shape[0], int(cin), int(cout), *out_shape[2:])
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    check(L.hno_conv3d_k3(ptr(x), ptr(W), ptr(bias), ptr(y), ptr(ws), nbytes, mode, x.shape[0], cin, cout, *x.shape[2:],
                          *out_shape[2:], stride, 1, act, stream_ptr()), 'hno_conv3d_k3')
    return y


class Conv3dK3Fn(_HnoFunction):
    """Conv3d(kernel 3, stride 1 or 2, padding 1) + bias as an implicit GEMM (V-Net-DS convolutions)."""

    @staticmethod
    def meta(x, W, bias, stride):
        return _m((x.shape[0], W.shape[0]) + tuple((v - 1) // stride + 1 for v in x.shape[2:]))

    @staticmethod
    def forward(ctx, x, W, bias, stride):
        x, W, bias = _f32c(x), _f32c(W), _f32c(bias)
        _need_gpu(x, W, bias)
        B, Cin = x.shape[:2]
        Cout = W.shape[0]
        osz = tuple((s - 1) // stride + 1 for s in x.shape[2:])
        y = _conv3d_call(x, W, bias, (B, Cout) + osz, 0, Cin, Cout, stride)
        ctx.save_for_backward(x, W)
        ctx.stride, ctx.has_bias = stride, bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = _f32c(g)
        B, Cin = x.shape[:2]
        Cout = W.shape[0]
        gx = _conv3d_call(g, W, None, tuple(x.shape), 1, Cin, Cout, ctx.stride) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W)
        ws = _conv_ws(Cin, Cout, True, x.device)
        check(_lib.lib().hno_conv3d_k3_wgrad(ptr(g), ptr(x), ptr(dW), ptr(ws), 0, B, Cin, Cout, *x.shape[2:], *g.shape[2:],
                                             ctx.stride, 1, stream_ptr()), 'hno_conv3d_k3_wgrad')
        return gx, dW, (_chan_sum(g) if ctx.has_bias else None), None


class ConvT3dK3Fn(_HnoFunction):
    """ConvTranspose3d(kernel 3, stride 2, padding 1, output_padding 1) + bias (V-Net-DS upsampling)."""

    @staticmethod
    def meta(x, Wt, bias):
        return _m((x.shape[0], Wt.shape[1]) + tuple(2 * v for v in x.shape[2:]))

    @staticmethod
    def forward(ctx, x, Wt, bias):
        x, Wt, bias = _f32c(x), _f32c(Wt), _f32c(bias)
        _need_gpu(x, Wt, bias)
        B, Cin = x.shape[:2]
        Cout = Wt.shape[1]
        osz = tuple(2 * s for s in x.shape[2:])
        y = _conv3d_call(x, Wt, bias, (B, Cout) + osz, 2, Cin, Cout, 2)
        ctx.save_for_backward(x, Wt)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, Wt = ctx.saved_tensors
        g = _f32c(g)
        B, Cin = x.shape[:2]
        Cout = Wt.shape[1]
        gx = _conv3d_call(g, Wt, None, tuple(x.shape), 3, Cin, Cout, 2) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(Wt)
        ws = _conv_ws(Cin, Cout, True, x.device)
        check(_lib.lib().hno_conv3d_k3_wgrad(ptr(g), ptr(x), ptr(dW), ptr(ws), 1, B, Cin, Cout, *x.shape[2:], *g.shape[2:],
                                             2, 1, stream_ptr()), 'hno_conv3d_k3_wgrad')
        return gx, dW, (_chan_sum(g) if ctx.has_bias else None)


class ConvKFn(_HnoFunction):
    """nn.Conv3d / nn.ConvTranspose3d of ConvNormAct / ConvTransposeNormAct (nets/nets_utils.py:136-211) with ANY kernel size
    (round 6: the reference's V-Net-DS takes ``kernel_size``, nets/architectures.py:55-70): the direct kernels hno_convk /
    hno_convk_wgrad.  ``pad`` None = k // 2 (stride 1 'same' for odd k, the strided and transposed forms of the reference); conv_in's
    Conv3d(k 2, s 2, p 1) beyond the fast kernel's channel limits takes pad = 1.  Transposed: stride 2, output_padding 1."""

    @staticmethod
    def _osz(spatial, k, stride, pad, transposed):
        if transposed:
            return tuple((v - 1) * 2 - 2 * pad + k + 1 for v in spatial)
        return tuple((v + 2 * pad - k) // stride + 1 for v in spatial)

    @staticmethod
    def meta(x, W, bias, stride, transposed, pad=None):
        k = int(W.shape[2])
        pad = k // 2 if pad is None else pad
        return _m((x.shape[0], W.shape[1] if transposed else W.shape[0]) + ConvKFn._osz(tuple(x.shape[2:]), k, stride, pad, transposed))

    @staticmethod
    def forward(ctx, x, W, bias, stride, transposed, pad=None):
        x, W, bias = _f32c(x), _f32c(W), _f32c(bias)
        _need_gpu(x, W, bias)
        k = int(W.shape[2])
        assert tuple(W.shape[2:]) == (k, k, k), 'cubic kernels'
        pad = k // 2 if pad is None else int(pad)
        B, Cin = x.shape[:2]
        Cout = W.shape[1] if transposed else W.shape[0]
        st = 2 if transposed else stride
        osz = ConvKFn._osz(tuple(x.shape[2:]), k, st, pad, transposed)
        y = torch.empty((B, Cout) + osz, device=x.device, dtype=torch.float32)
        check(_lib.lib().hno_convk(ptr(x), ptr(W), ptr(bias), ptr(y), 2 if transposed else 0, B, Cin, Cout, *x.shape[2:], *osz, k, st, pad,
                                   stream_ptr()), 'hno_convk')
        ctx.save_for_backward(x, W)
        ctx.cfg = (k, st, pad, bool(transposed), bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        k, st, pad, transposed, has_bias = ctx.cfg
        g = _f32c(g)
        B, Cin = x.shape[:2]
        Cout = W.shape[1] if transposed else W.shape[0]
        L = _lib.lib()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            check(L.hno_convk(ptr(g), ptr(W), None, ptr(gx), 3 if transposed else 1, B, Cin, Cout, *g.shape[2:], *x.shape[2:], k, st, pad,
                              stream_ptr()), 'hno_convk')
        dW = torch.empty_like(W)
        check(L.hno_convk_wgrad(ptr(g), ptr(x), ptr(dW), 1 if transposed else 0, B, Cin, Cout, *x.shape[2:], *g.shape[2:], k, st, pad,
                                stream_ptr()), 'hno_convk_wgrad')
        return gx, dW, (_chan_sum(g) if has_bias else None), None, None, None


class GroupNormActFn(_HnoFunction):
    """act(GroupNorm(1, C)(x)) (nets/nets_utils.py:127-133 with :165-170)."""

    @staticmethod
    def meta(x, gamma, beta, eps, act):
        return _m(x.shape)

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act):
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        _need_gpu(x, gamma, beta)
        B, C = x.shape[:2]
        V = _flat_v(x)
        y = torch.empty_like(x)
        mr = torch.empty((B, 2), device=x.device, dtype=torch.float32)
        ws = torch.empty(2 * B, device=x.device, dtype=torch.float64)
        check(_lib.lib().hno_groupnorm1_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mr), ptr(ws), B, C, V, float(eps), act,
                                            stream_ptr()), 'hno_groupnorm1_fwd')
        ctx.save_for_backward(x, y, mr, gamma)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, mr, gamma = ctx.saved_tensors
        g = _f32c(g)
        B, C = x.shape[:2]
        V = _flat_v(x)
        gx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        sums = torch.empty(2 * B * C, device=x.device, dtype=torch.float64)
        coef = torch.empty(2 * B, device=x.device, dtype=torch.float32)
        check(_lib.lib().hno_groupnorm1_bwd(ptr(g), ptr(y), ptr(x), ptr(mr), ptr(gamma), ptr(gx), ptr(dg), ptr(db), ptr(sums),
                                            ptr(coef), B, C, V, ctx.act, stream_ptr()), 'hno_groupnorm1_bwd')
        return gx, dg, db, None, None


class NearestUpFn(_HnoFunction):
    """F.interpolate(x, size) with the default nearest mode (`upsampling`, nets/architectures.py:638-653)."""

    @staticmethod
    def meta(x, size):
        return _m(tuple(x.shape[:2]) + tuple(size))

    @staticmethod
    def forward(ctx, x, size):
        x = _f32c(x)
        _need_gpu(x)
        B, C, d, h, w = x.shape
        D, H, W = (int(s) for s in size)
        y = torch.empty((B, C, D, H, W), device=x.device, dtype=torch.float32)
        check(_lib.lib().hno_nearest3d(ptr(x), ptr(y), B * C, d, h, w, D, H, W, 0, 0, stream_ptr()), 'hno_nearest3d')
        ctx.lr = (B, C, d, h, w)
        return y

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)
        B, C, d, h, w = ctx.lr
        gx = torch.empty(ctx.lr, device=g.device, dtype=torch.float32)
        check(_lib.lib().hno_nearest3d(ptr(g), ptr(gx), B * C, d, h, w, *g.shape[2:], 1, 0, stream_ptr()), 'hno_nearest3d')
        return gx, None


class PerModeHartleyFn(_HnoFunction):
    """hartley_conv with per-mode weights (nets/hartley_operator.py:302-317): x, xr (B,Ci,d0,d1,d2), w (Co,Ci,d0,d1,d2)."""

    @staticmethod
    def meta(x, xr, w):
        return _m((x.shape[0], w.shape[0]) + tuple(w.shape[2:]))

    @staticmethod
    def forward(ctx, x, xr, w):
        x, xr, w = _f32c(x), _f32c(xr), _f32c(w)
        _need_gpu(x, xr, w)
        B, Ci = x.shape[:2]
        Co, (d0, d1, d2) = w.shape[0], w.shape[2:]
        M = d0 * d1 * d2
        y = torch.empty((B, Co, d0, d1, d2), device=x.device, dtype=torch.float32)
        check(_lib.lib().hno_permode_fwd(ptr(x), ptr(xr), ptr(w), None, ptr(y), B, Ci, Co, M, d0, d1, d2, 0, stream_ptr()),
              'hno_permode_fwd')
        ctx.save_for_backward(x, xr, w)
        return y

    @staticmethod
    def backward(ctx, g):
        x, xr, w = ctx.saved_tensors
        g = _f32c(g)
        B, Ci = x.shape[:2]
        Co, (d0, d1, d2) = w.shape[0], w.shape[2:]
        gx, gxr, dw = torch.empty_like(x), torch.empty_like(xr), torch.empty_like(w)
        check(_lib.lib().hno_permode_bwd(ptr(g), ptr(x), ptr(xr), ptr(w), None, ptr(gx), ptr(gxr), ptr(dw), None, B, Ci, Co,
                                         d0 * d1 * d2, d0, d1, d2, 0, stream_ptr()), 'hno_permode_bwd')
        return gx, gxr, dw


class PerModeFourierFn(_HnoFunction):
    """Complex per-mode mix 'oidhw,bidhw->bodhw' (nets/fourier_operator.py:174-191) on [re | im] planes:
    spec (B, 2Ci, ...), wr / wi (Co, Ci, 2m0, 2m1, m2) -> (B, 2Co, ...)."""

    @staticmethod
    def meta(spec, wr, wi):
        return _m((spec.shape[0], 2 * wr.shape[0]) + tuple(spec.shape[2:]))

    @staticmethod
    def forward(ctx, spec, wr, wi):
        spec, wr, wi = _f32c(spec), _f32c(wr), _f32c(wi)
        _need_gpu(spec, wr, wi)
        B, Ci = spec.shape[0], spec.shape[1] // 2
        Co = wr.shape[0]
        M = int(np.prod(wr.shape[2:]))
        y = torch.empty((B, 2 * Co) + tuple(spec.shape[2:]), device=spec.device, dtype=torch.float32)
        check(_lib.lib().hno_permode_fwd(ptr(spec), None, ptr(wr), ptr(wi), ptr(y), B, Ci, Co, M, 1, 1, M, 1, stream_ptr()),
              'hno_permode_fwd')
        ctx.save_for_backward(spec, wr, wi)
        return y

    @staticmethod
    def backward(ctx, g):
        spec, wr, wi = ctx.saved_tensors
        g = _f32c(g)
        B, Ci = spec.shape[0], spec.shape[1] // 2
        Co = wr.shape[0]
        M = int(np.prod(wr.shape[2:]))
        gx, dwr, dwi = torch.empty_like(spec), torch.empty_like(wr), torch.empty_like(wi)
        check(_lib.lib().hno_permode_bwd(ptr(g), ptr(spec), None, ptr(wr), ptr(wi), ptr(gx), None, ptr(dwr), ptr(dwi), B, Ci,
                                         Co, M, 1, 1, M, 1, stream_ptr()), 'hno_permode_bwd')
        return gx, dwr, dwi


def bmm_raw(A, B, transA, transB, alpha=1.0):
    """(batch..., M|K, K|M) x (batch..., K|N, N|K) -> (batch..., M, N) on the fp32 matrix cores."""
    _need_gpu(A, B)
    lead = A.shape[:-2]
    batch = int(np.prod(lead)) if lead else 1
    M, K = (A.shape[-1], A.shape[-2]) if transA else (A.shape[-2], A.shape[-1])
    N = B.shape[-2] if transB else B.shape[-1]
    assert (B.shape[-1] if transB else B.shape[-2]) == K and B.shape[:-2] == lead
    C = torch.empty(tuple(lead) + (M, N), device=A.device, dtype=torch.float32)
    check(_lib.lib().hno_bmm(ptr(A), ptr(B), ptr(C), batch, M, N, K, int(transA), int(transB), float(alpha), stream_ptr()),
          'hno_bmm')
    return C


class BmmFn(_HnoFunction):
    """C = alpha * op(A) op(B), batched (nets/hartley_mha.py:196-201)."""

    @staticmethod
    def meta(A, B, transA, transB, alpha):
        M = A.shape[-1] if transA else A.shape[-2]
        N = B.shape[-2] if transB else B.shape[-1]
        return _m(tuple(A.shape[:-2]) + (M, N))

    @staticmethod
    def forward(ctx, A, B, transA, transB, alpha):
        A, B = _f32c(A), _f32c(B)
        ctx.save_for_backward(A, B)
        ctx.cfg = (bool(transA), bool(transB), float(alpha))
        return bmm_raw(A, B, transA, transB, alpha)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        tA, tB, alpha = ctx.cfg
        g = _f32c(g)
        # C = op(A) op(B): d op(A) = g op(B)^T, d op(B) = op(A)^T g; transpose back where op is a transpose
        if not tA:
            dA = bmm_raw(g, B, False, not tB, alpha)            # (M,N) x op(B)^T -> (M,K)
        else:
            dA = bmm_raw(B, g, tB, True, alpha)                 # op(B) g^T -> (K,M)
        if not tB:
            dB = bmm_raw(A, g, not tA, False, alpha)            # op(A)^T g -> (K,N)
        else:
            dB = bmm_raw(g, A, True, tA, alpha)                 # g^T op(A) -> (N,K)
        return dA, dB, None, None, None


class HartleyAttentionFn(_HnoFunction):
    """out = V att^T, att = act(alpha Q^T K), per (batch, head): q, k (B, Z, Ck, T), v (B, Z, Cv, T) -> (B, Z, Cv, T)
    (nets/hartley_mha.py:196-201).  One fused launch each way (hno_hmha_fwd / hno_hmha_bwd): the (B, Z, T, T) attention
    matrix is never written, the backward recomputes it; only q, k, v are saved."""

    @staticmethod
    def meta(q, k, v, alpha, act):
        return _m(v.shape)

    @staticmethod
    def forward(ctx, q, k, v, alpha, act):
        q, k, v = _f32c(q), _f32c(k), _f32c(v)
        _need_gpu(q, k, v)
        B, Z, Ck, T = q.shape
        Cv = v.shape[2]
        assert k.shape == q.shape and v.shape[:2] == q.shape[:2] and v.shape[3] == T
        out = torch.empty_like(v)
        ws = torch.empty(_lib.lib().hno_hmha_workspace_bytes(B * Z, Ck, Cv, T) // 4, device=q.device, dtype=torch.float32)
        check(_lib.lib().hno_hmha_fwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(ws), 4 * ws.numel(), B * Z, Ck, Cv, T, float(alpha), act,
                                      stream_ptr()), 'hno_hmha_fwd')
        ctx.save_for_backward(q, k, v)
        ctx.cfg = (float(alpha), act)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v = ctx.saved_tensors
        alpha, act = ctx.cfg
        g = _f32c(g)
        B, Z, Ck, T = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(_lib.lib().hno_hmha_workspace_bytes(B * Z, Ck, v.shape[2], T) // 4, device=q.device, dtype=torch.float32)
        check(_lib.lib().hno_hmha_bwd(ptr(q), ptr(k), ptr(v), ptr(g), ptr(dq), ptr(dk), ptr(dv), ptr(ws), 4 * ws.numel(), B * Z, Ck,
                                      v.shape[2], T, alpha, act, stream_ptr()), 'hno_hmha_bwd')
        return dq, dk, dv, None, None


class PatchGroupQKVFn(_HnoFunction):
    """The stacked q / k / v projections (B, Z Kq + Z Kk + Z Kv, d, h, w) -> three contiguous (B, Z, K P, T) attention operands:
    reference's split + reshape + grouping3d (nets/hartley_mha.py:180-190, 473-498) as ONE permutation launch each way
    (hno_patch_group3) instead of ~10 ATen copies forward and as many plus a concatenation backward."""

    @staticmethod
    def meta(y, Z, Kq, Kk, Kv, patch):
        B, _, d, h, w = y.shape
        P = int(np.prod(patch))
        T = (d // patch[0]) * (h // patch[1]) * (w // patch[2])
        return tuple(_m((B, Z, K * P, T)) for K in (Kq, Kk, Kv))

    @staticmethod
    def forward(ctx, y, Z, Kq, Kk, Kv, patch):
        y = _f32c(y)
        _need_gpu(y)
        B, Ct, d, h, w = y.shape
        assert Ct == Z * (Kq + Kk + Kv)
        pd, ph, pw = (int(v) for v in patch)
        P, T = pd * ph * pw, (d // pd) * (h // ph) * (w // pw)
        outs = [torch.empty((B, Z, K * P, T), device=y.device, dtype=torch.float32) for K in (Kq, Kk, Kv)]
        check(_lib.lib().hno_patch_group3(ptr(y), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), B, Z * Kq, Z * Kk, Z * Kv, d, h, w, pd, ph, pw, 0,
                                          stream_ptr()), 'hno_patch_group3')
        ctx.cfg = (tuple(y.shape), Z, Kq, Kk, Kv, (pd, ph, pw))
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, gq, gk, gv):
        shape, Z, Kq, Kk, Kv, (pd, ph, pw) = ctx.cfg
        B, Ct, d, h, w = shape
        gs = [None if g is None else _f32c(g) for g in (gq, gk, gv)]
        dev = next(g for g in gs if g is not None).device
        gy = torch.empty(shape, device=dev, dtype=torch.float32)
        check(_lib.lib().hno_patch_group3(ptr(gy), ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), B, Z * Kq, Z * Kk, Z * Kv, d, h, w, pd, ph, pw, 1,
                                          stream_ptr()), 'hno_patch_group3')
        return gy, None, None, None, None, None


class PatchUngroupFn(_HnoFunction):
    """ungrouping3d of the attention output (B, Z, V P, T) -> (B, Z V, d, h, w) (nets/hartley_mha.py:501-524) as one launch each way."""

    @staticmethod
    def meta(x, Z, V, patch, grid):
        return _m((x.shape[0], Z * V) + tuple(grid))

    @staticmethod
    def forward(ctx, x, Z, V, patch, grid):
        x = _f32c(x)
        _need_gpu(x)
        B = x.shape[0]
        d, h, w = (int(v) for v in grid)
        pd, ph, pw = (int(v) for v in patch)
        out = torch.empty((B, Z * V, d, h, w), device=x.device, dtype=torch.float32)
        check(_lib.lib().hno_patch_group3(ptr(out), ptr(x), None, None, B, Z * V, 0, 0, d, h, w, pd, ph, pw, 1, stream_ptr()), 'hno_patch_group3')
        ctx.cfg = (tuple(x.shape), Z, V, (pd, ph, pw), (d, h, w))
        return out

    @staticmethod
    def backward(ctx, g):
        shape, Z, V, (pd, ph, pw), (d, h, w) = ctx.cfg
        g = _f32c(g)
        gx = torch.empty(shape, device=g.device, dtype=torch.float32)
        check(_lib.lib().hno_patch_group3(ptr(g), ptr(gx), None, None, shape[0], Z * V, 0, 0, d, h, w, pd, ph, pw, 0, stream_ptr()), 'hno_patch_group3')
        return gx, None, None, None, None


class GroupedAttentionFn(_HnoFunction):
    """Stacked q / k / v projections (B, Z (2 Kq + Kv), d, h, w) -> attention output (B, Z Kv, d, h, w): grouping3d, the fused attention
    and ungrouping3d (nets/hartley_mha.py:180-216, 473-524) as one autograd node.  The attention kernels leave the partial results of
    their stream splits unsummed (hno_hmha_fwd_parts / _bwd_parts) and the ungrouping permutation adds them in slice order while it
    reads them (hno_patch_group3_sum): 3 launches forward and 4 backward per block instead of 4 + 7 (and ~40 as torch ops)."""

    @staticmethod
    def meta(y, Z, Kq, Kv, patch, alpha, act):
        return _m((y.shape[0], Z * Kv) + tuple(y.shape[2:]))

    @staticmethod
    def supported(Z, Kq, Kv, patch, act):
        P = int(np.prod(patch))
        return bool(_lib.lib().hno_hmha_parts_supported(Kq * P, Kv * P, act))

    @staticmethod
    def forward(ctx, y, Z, Kq, Kv, patch, alpha, act):
        y = _f32c(y)
        _need_gpu(y)
        L = _lib.lib()
        B, Ct, d, h, w = y.shape
        assert Ct == Z * (2 * Kq + Kv)
        pd, ph, pw = (int(v) for v in patch)
        P, T = pd * ph * pw, (d // pd) * (h // ph) * (w // pw)
        q, k = (torch.empty((B, Z, Kq * P, T), device=y.device, dtype=torch.float32) for _ in range(2))
        v = torch.empty((B, Z, Kv * P, T), device=y.device, dtype=torch.float32)
        check(L.hno_patch_group3(ptr(y), ptr(q), ptr(k), ptr(v), B, Z * Kq, Z * Kq, Z * Kv, d, h, w, pd, ph, pw, 0, stream_ptr()), 'hno_patch_group3')
        ns = L.hno_hmha_nsplit(B * Z, T)
        parts = torch.empty((ns, B, Z, Kv * P, T), device=y.device, dtype=torch.float32)
        check(L.hno_hmha_fwd_parts(ptr(q), ptr(k), ptr(v), ptr(parts), B * Z, Kq * P, Kv * P, T, float(alpha), act, stream_ptr()), 'hno_hmha_fwd_parts')
        out = torch.empty((B, Z * Kv, d, h, w), device=y.device, dtype=torch.float32)
        check(L.hno_patch_group3_sum(ptr(out), ptr(parts), None, None, ns, B, Z * Kv, 0, 0, d, h, w, pd, ph, pw, stream_ptr()), 'hno_patch_group3_sum')
        ctx.save_for_backward(q, k, v)
        ctx.cfg = (tuple(y.shape), Z, Kq, Kv, (pd, ph, pw), float(alpha), act, ns)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v = ctx.saved_tensors
        shape, Z, Kq, Kv, (pd, ph, pw), alpha, act, _ = ctx.cfg
        B, Ct, d, h, w = shape
        P, T = pd * ph * pw, q.shape[3]
        L = _lib.lib()
        ns = L.hno_hmha_nsplit_bwd(B * Z, Kq * P, Kv * P, T)      # (the backward kernels may take fewer stream splits than the forward one)
        g = _f32c(g)
        dout = torch.empty_like(v)
        check(L.hno_patch_group3(ptr(g), ptr(dout), None, None, B, Z * Kv, 0, 0, d, h, w, pd, ph, pw, 0, stream_ptr()), 'hno_patch_group3')
        dq, dk = (torch.empty((ns,) + tuple(q.shape), device=g.device, dtype=torch.float32) for _ in range(2))
        dv = torch.empty((ns,) + tuple(v.shape), device=g.device, dtype=torch.float32)
        check(L.hno_hmha_bwd_parts(ptr(q), ptr(k), ptr(v), ptr(dout), ptr(dq), ptr(dk), ptr(dv), B * Z, Kq * P, Kv * P, T, alpha, act, stream_ptr()),
              'hno_hmha_bwd_parts')
        gy = torch.empty(shape, device=g.device, dtype=torch.float32)
        check(L.hno_patch_group3_sum(ptr(gy), ptr(dq), ptr(dk), ptr(dv), ns, B, Z * Kq, Z * Kq, Z * Kv, d, h, w, pd, ph, pw, stream_ptr()),
              'hno_patch_group3_sum')
        return gy, None, None, None, None, None, None


def hmha_supported(Ck, Cv):
    return bool(_lib.lib().hno_hmha_supported(int(Ck), int(Cv)))


class ActFn(_HnoFunction):

    @staticmethod
    def meta(x, act):
        return _m(x.shape)
    @staticmethod
    def forward(ctx, x, act):
        x = _f32c(x)
        _need_gpu(x)
        y = torch.empty_like(x)
        check(_lib.lib().hno_act_fwd(ptr(x), ptr(y), x.numel(), act, stream_ptr()), 'hno_act_fwd')
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return act_bwd_raw(_f32c(g), y, ctx.act), None


class BiasActFn(_HnoFunction):
    """act(x + bias[c]) for a per-channel bias (hno_bias_act; the reference adds a (1, Co, 1, 1, 1) parameter to the spectrum and applies
    SELU, nets/hartley_operator.py:262-267)."""

    @staticmethod
    def meta(x, bias, act):
        return _m(x.shape)

    @staticmethod
    def forward(ctx, x, bias, act):
        _need_gpu(x, bias)
        y = _f32c(x).clone()
        b = _f32c(bias.reshape(-1))
        assert b.numel() == y.shape[1], 'one bias per channel'
        check(_lib.lib().hno_bias_act(ptr(y), ptr(b), y.shape[0], y.shape[1], _flat_v(y), act, stream_ptr()), 'hno_bias_act')
        ctx.act, ctx.bshape = act, tuple(bias.shape)
        ctx.save_for_backward(y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _f32c(g)
        if ctx.act != ACT_NONE:
            g = act_bwd_raw(g, y, ctx.act)
        db = _chan_sum(g).reshape(ctx.bshape) if ctx.needs_input_grad[1] else None
        return g, db, None


class AddFn(_HnoFunction):

    @staticmethod
    def meta(a, b):
        return _m(a.shape)
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        _need_gpu(a, b)
        out = torch.empty_like(a)
        check(_lib.lib().hno_add(ptr(a), ptr(b), ptr(out), a.numel(), stream_ptr()), 'hno_add')
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class AxpbyFn(_HnoFunction):
    """alpha * a + beta * b (hno_axpby)."""

    @staticmethod
    def meta(a, b, alpha, beta):
        return _m(a.shape)

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        a, b = _f32c(a), _f32c(b)
        _need_gpu(a, b)
        out = torch.empty_like(a)
        check(_lib.lib().hno_axpby(float(alpha), ptr(a), float(beta), ptr(b), ptr(out), a.numel(), stream_ptr()), 'hno_axpby')
        ctx.ab = (float(alpha), float(beta))
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)

        def scaled(c):
            if c == 1.0:
                return g
            out = torch.empty_like(g)
            check(_lib.lib().hno_axpby(c, ptr(g), 0.0, None, ptr(out), g.numel(), stream_ptr()), 'hno_axpby')
            return out
        return (scaled(ctx.ab[0]) if ctx.needs_input_grad[0] else None,
                scaled(ctx.ab[1]) if ctx.needs_input_grad[1] else None, None, None)


class SpecMixFn(_HnoFunction):
    """L stacked shared-weight frequency-domain mixes z <- act(W z + residual z)
    (nets/hnosegxs.py:307-329, nets/hartley_operator.py:287-292).  The layer weights are separate
    (C, C) tensors (one Parameter per layer, as in the reference's state dict)."""

    @staticmethod
    def meta(z0, residual, act, *Ws):
        return _m(z0.shape)

    @staticmethod
    def forward(ctx, z0, residual, act, *Ws):
        z0 = _f32c(z0)
        Ws = [_f32c(w) for w in Ws]
        _need_gpu(z0, *Ws)
        zs = specmix_fwd_raw(z0, Ws, residual, act)
        ctx.save_for_backward(z0, zs, *Ws)
        ctx.residual, ctx.act = int(residual), act
        return zs[-1]

    @staticmethod
    def backward(ctx, g):
        z0, zs, *Ws = ctx.saved_tensors
        gz0, dW = specmix_bwd_raw(_f32c(g), z0, zs, Ws, ctx.residual, ctx.act)
        return (gz0, None, None) + tuple(dW.unbind(0))


class PwConvFn(_HnoFunction):
    """act(W [xa ; xb] + bias): fused concat + 1x1x1 conv + bias + activation
    (nets/hnosegxs.py:274-275; nets/nets_utils.py:127-133)."""

    @staticmethod
    def meta(xa, xb, W, bias, act):
        return _m((xa.shape[0], W.shape[0]) + tuple(xa.shape[2:]))

    @staticmethod
    def _wide(xa, xb, W):
        """Wide layers on small grids (deep V-Net levels: 96 ... 384 channels, <= 16 K voxels) are plain GEMMs: the
        streaming pointwise kernels would need (Cout / 32) x (Cin / 64) launches, the LDS-tiled batched GEMM needs one."""
        launches = ((W.shape[0] + 31) // 32) * ((xa.shape[1] + 63) // 64)   # what the streaming path would need
        # the GEMM's weight gradient reduces over V inside one tile (no split-K): only worth it when Cin is large too
        return xb is None and launches >= 6 and xa.shape[1] >= 64 and _flat_v(xa) <= 16384

    @staticmethod
    def forward(ctx, xa, xb, W, bias, act):
        ctx.leaf_params = _leaf_params(ctx, W, bias)
        xa, xb, W, bias = _f32a(xa), _f32a(xb), _f32c(W), _f32c(bias)
        _need_gpu(xa, xb, W, bias)
        ctx.wide = PwConvFn._wide(xa, xb, W)
        if ctx.wide:
            xa = _f32c(xa)
            B, Cin = xa.shape[:2]
            Cout, V = W.shape[0], _flat_v(xa)
            w2 = W.reshape(1, Cout, Cin).expand(B, Cout, Cin).contiguous() if B > 1 else W.reshape(1, Cout, Cin)
            y = bmm_raw(w2, xa.reshape(B, Cin, V), False, False).reshape((B, Cout) + tuple(xa.shape[2:]))
            if bias is not None or act != ACT_NONE:
                check(_lib.lib().hno_bias_act(ptr(y), ptr(bias), B, Cout, V, act, stream_ptr()), 'hno_bias_act')
        else:
            ctx.bf16 = _autocast_bf16()
            y = pwconv_fwd_raw(xa, xb, W, bias, act, ctx.bf16)
        ctx.save_for_backward(xa, xb, W, y if act != ACT_NONE else None, bias)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        xa, xb, W, y, bias = ctx.saved_tensors
        if ctx.wide:
            g = _f32c(gy)
            if ctx.act != ACT_NONE:
                g = act_bwd_raw(g, y, ctx.act)
            B, Cin = xa.shape[:2]
            Cout, V = W.shape[0], _flat_v(xa)
            g3, x3 = g.reshape(B, Cout, V), xa.reshape(B, Cin, V)
            gxa = None
            if ctx.needs_input_grad[0]:
                w2 = W.reshape(1, Cout, Cin).expand(B, Cout, Cin).contiguous() if B > 1 else W.reshape(1, Cout, Cin)
                gxa = bmm_raw(w2, g3, True, False).reshape(xa.shape)
            dW = bmm_raw(g3, x3, False, True)                     # (B, Cout, Cin)
            acc = dW[0]
            for bi in range(1, B):       # batch > 1: fixed-order sum of the per-sample products (hno_add)
                nxt = torch.empty_like(acc)
                check(_lib.lib().hno_add(ptr(acc), ptr(dW[bi]), ptr(nxt), acc.numel(), stream_ptr()), 'hno_add')
                acc = nxt
            dW = acc.reshape(W.shape)
            db = _chan_sum(g) if ctx.has_bias else None
            return gxa, None, dW, db, None
        gxa, gxb, dW, db = pwconv_bwd_raw(_f32a(gy), y, xa, xb, W, ctx.act, ctx.has_bias, ctx.needs_input_grad[0],
                                          ctx.needs_input_grad[1], defer=ctx.leaf_params and _release_use(ctx, W, bias) and _deferrable(W, bias),
                                          bias=bias, bf16=getattr(ctx, 'bf16', False))
        return gxa, gxb, dW, db, None


class MultiPwConvFn(_HnoFunction):
    """bias + sum_t W[t] x_t over T equally wide tensors: torch.cat(tensors, dim=1) + a k = 1 convolution (the reference's
    deep-supervision head, nets/architectures.py:341-343) as ONE launch each way (hno_pwmulti_fwd / hno_pwmulti_bwd).

        forward(w_tkc (T, K, C), bias (K,) or None, *tensors (B, C, ...)) -> (B, K, ...)"""

    @staticmethod
    def supported(tensors, K):
        t0 = tensors[0]
        if t0.is_meta:
            return True
        if not (t0.is_cuda and t0.dtype == torch.float32 and t0.ndim == 5):
            return False
        ld = chan_stride(t0)
        same = all(t.shape == t0.shape and t.dtype == t0.dtype and t.device == t0.device and chan_stride(t) == ld
                   and (ld is not None or t.is_contiguous()) for t in tensors)
        return bool(same and _lib.lib().hno_pwmulti_supported(len(tensors), t0.shape[1], K))

    @staticmethod
    def meta(w, bias, *tensors):
        return _m((tensors[0].shape[0], w.shape[1]) + tuple(tensors[0].shape[2:]))

    @staticmethod
    def forward(ctx, w, bias, *tensors):
        import ctypes
        _need_gpu(w, bias, *tensors)
        w, bias = _f32c(w), _f32c(bias)
        T, K, C = w.shape
        t0 = tensors[0]
        assert len(tensors) == T and t0.shape[1] == C
        ld = chan_stride(t0)
        V = ld or _flat_v(t0)           # channel-padded operands: the kernel runs over V := ld voxels per channel (pwconv_fwd_raw)
        out = act_like(t0, K)
        xs = (ctypes.c_void_p * T)(*[t.data_ptr() for t in tensors])
        check(_lib.lib().hno_pwmulti_fwd(xs, T, C, ptr(w), ptr(bias), ptr(out), t0.shape[0], K, V, V, stream_ptr()), 'hno_pwmulti_fwd')
        ctx.save_for_backward(w, *tensors)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        w, *tensors = ctx.saved_tensors
        T, K, C = w.shape
        t0 = tensors[0]
        ld = chan_stride(t0)
        V, B = ld or _flat_v(t0), t0.shape[0]          # (g's padding is zero: to_layout)
        g = to_layout(_f32a(g), ld)
        L = _lib.lib()
        gxs = [act_like(t) if ctx.needs_input_grad[2 + i] else None for i, t in enumerate(tensors)]
        dW = torch.empty_like(w)
        db = torch.empty(K, device=w.device, dtype=torch.float32) if ctx.has_bias else None
        nws = L.hno_pwmulti_bwd_workspace_bytes(T, C, K, B, V)
        ws = torch.empty(max(nws // 4, 4), device=w.device, dtype=torch.float32)
        xs = (ctypes.c_void_p * T)(*[t.data_ptr() for t in tensors])
        gp = (ctypes.c_void_p * T)(*[(None if t is None else t.data_ptr()) for t in gxs])
        check(L.hno_pwmulti_bwd(ptr(g), xs, gp, T, C, ptr(w), ptr(dW), ptr(db), ptr(ws), nws, B, K, V, V, stream_ptr()), 'hno_pwmulti_bwd')
        return (dW, db) + tuple(gxs)


class ComplexMixFn(_HnoFunction):
    """Complex shared-weight channel mix on the [re | im] layout (nets/fourier_operator.py:164-172):
    spec (B, 2Ci, ...) -> (B, 2Co, ...) with W = weight_real + i weight_imag, as one real pointwise conv with the
    composed matrix [[Wr, -Wi], [Wi, Wr]] (built and split back by two tiny kernels: no ATen cat / neg / slice)."""

    @staticmethod
    def meta(spec, wr, wi):
        return _m((spec.shape[0], 2 * wr.shape[0]) + tuple(spec.shape[2:]))

    @staticmethod
    def forward(ctx, spec, wr, wi):
        spec, wr, wi = _f32c(spec), _f32c(wr), _f32c(wi)
        _need_gpu(spec, wr, wi)
        Co, Ci = wr.shape
        assert spec.shape[1] == 2 * Ci and wi.shape == wr.shape
        w2 = torch.empty((2 * Co, 2 * Ci), device=spec.device, dtype=torch.float32)
        check(_lib.lib().hno_cmix_compose(ptr(wr), ptr(wi), ptr(w2), Co, Ci, stream_ptr()), 'hno_cmix_compose')
        y = pwconv_fwd_raw(spec, None, w2, None, ACT_NONE)
        ctx.save_for_backward(spec, w2)
        return y

    @staticmethod
    def backward(ctx, g):
        spec, w2 = ctx.saved_tensors
        gx, _, dw2, _ = pwconv_bwd_raw(_f32c(g), None, spec, None, w2, ACT_NONE, False, ctx.needs_input_grad[0], False)
        Co, Ci = w2.shape[0] // 2, w2.shape[1] // 2
        dwr = torch.empty((Co, Ci), device=g.device, dtype=torch.float32)
        dwi = torch.empty_like(dwr)
        check(_lib.lib().hno_cmix_split_grad(ptr(dw2), ptr(dwr), ptr(dwi), Co, Ci, stream_ptr()), 'hno_cmix_split_grad')
        return gx, dwr, dwi


def noblock_io16_ok(spatial, channels, fourier, modes, br_shape, cat_shape):
    """the block can keep its input and output as bf16 tensors (autocast): the Fourier block of FNOSeg (24 channels, conv branch, concat
    skip) on a working grid whose planes have a bf16 item kernel (65 x 65) and the fused Fourier middle"""
    if not (io16_enabled() and fourier and len(spatial) == 3 and channels == 24 and tuple(br_shape or ()) [:2] == (24, 24)
            and tuple(cat_shape)[:2] == (24, 48) and tuple(spatial[1:]) == (65, 65) and int(np.prod(spatial)) % 32 != 0):
        return False
    m = clamp_modes(tuple(modes), tuple(spatial))
    return (os.environ.get('HNO_FUSED_MID', '1') != '0' and os.environ.get('HNO_FUSED_MID_BWD', '1') != '0'
            and bool(_lib.lib().hno_spec_mid_fourier_supported(24, int(spatial[0]), int(m[0]), int(m[1]), int(m[2]))))


class CastFn(_HnoFunction):
    """fp32 <-> bf16 of a channel-padded activation, layout kept: the two ends of a chain of blocks with bf16 activations in memory --
    what autocast's to(bfloat16) in front of the first nn.Conv3d, and type promotion behind the last one, do in the reference
    (experiments/train_test.py:154-160).  The gradient takes the opposite cast."""

    @staticmethod
    def meta(x, to_bf16):
        return _m(x.shape, torch.bfloat16 if to_bf16 else torch.float32)

    @staticmethod
    def _cast(x, to_bf16):
        if to_bf16:
            ld = chan_stride(x) if x.dtype == torch.float32 else chan_stride16(x)
            if ld is None:
                x = to_layout(x.float(), _pad_ld(_flat_v(x)))
                ld = chan_stride(x)
            return to_bf16_layout(x, ld)
        ld = chan_stride16(x)
        if ld is None:
            return _f32c(x)
        out = act_empty(x.shape[0], x.shape[1], x.shape[2:], x.device, ld)
        check(_lib.lib().hno_cast_bf16_f32(ptr(x), ptr(out), x.shape[0] * x.shape[1] * ld, stream_ptr()), 'hno_cast_bf16_f32')
        return out

    @staticmethod
    def forward(ctx, x, to_bf16):
        _need_gpu(x)
        ctx.to_bf16 = bool(to_bf16)
        return CastFn._cast(x, ctx.to_bf16)

    @staticmethod
    def backward(ctx, g):
        return CastFn._cast(g, not ctx.to_bf16), None


class NOBlockFn(_HnoFunction):
    """One FNOSeg / HNOSeg block (nets/architectures.py:511-608 with shared weights, SELU, concat skip) as a single
    autograd node:

        y   = act(op(x) + conv_branch(x))          op = Fourier / Hartley operator with use_transform (crop, mix, pad)
        out = act(conv_concat(cat[y, x]))

    The block input x has three consumers (operator, conv branch, concat skip).  Owning the block lets the backward
    accumulate the three gradients inside the kernels that produce them -- the branch conv accumulates into the
    concat-path gradient, the inverse transform adds that sum on its store -- and take the activation gradient of y
    inside the forward transform and the branch conv instead of a separate elementwise pass."""

    @staticmethod
    def meta(x, fourier, modes, act, br_w, br_b, cat_w, cat_b, *op_ws):
        return _m((x.shape[0], cat_w.shape[0]) + tuple(x.shape[2:]))

    @staticmethod
    def forward(ctx, x, fourier, modes, act, br_w, br_b, cat_w, cat_b, *op_ws):
        # op_ws: (weight,) of a Hartley operator; (weight_real, weight_imag[, w2]) of a Fourier operator -- w2: their composed real form
        # when the model built it for all its blocks at once (ops.cmix_compose_all; not a parameter: no gradient)
        w_pre = op_ws[2] if (fourier and len(op_ws) == 3) else None
        ctx.n_ops = len(op_ws)
        op_ws = op_ws[:2] if fourier else op_ws
        ctx.leaf_params = _leaf_params(ctx, br_w, br_b, cat_w, cat_b, *op_ws)
        br_w, br_b, cat_w, cat_b = (_f32c(t) for t in (br_w, br_b, cat_w, cat_b))
        op_ws = [_f32c(w) for w in op_ws]
        spatial = tuple(x.shape[2:])
        if len(modes) == 2:                      # 2-D model on a (B, C, 1, H, W) view
            modes = (0,) + tuple(modes)
        modes = clamp_modes(modes, spatial)
        bf = _autocast_bf16()       # autocast: the spatial 1x1x1 convolutions take bf16 operands; transform and spectral mix stay fp32
        # bf16 activations in memory (round 6): a channel-padded bf16 input stays bf16, and so does the block's output
        io16 = (bf and chan_stride16(x) is not None and
                noblock_io16_ok(spatial, x.shape[1], fourier, modes, None if br_w is None else br_w.shape, cat_w.shape))
        if not io16:
            # a channel-padded input stays padded through the block when both transform directions take the stride (XSBlockFn)
            x = _f32a(x) if (chan_stride(x) is not None and padded_ok(spatial, modes)) else _f32c(x)
        ld = chan_stride16(x) if io16 else chan_stride(x)
        _need_gpu(x, cat_w, *op_ws)
        n3 = float(np.prod(spatial))
        # 24 + 24 -> 24 with a conv branch: branch conv, add, activation and concat conv in one pass after the inverse
        fuse_tail = br_w is not None and tuple(cat_w.shape[:2]) == (24, 48) and tuple(br_w.shape[:2]) == (24, 24)
        x2 = pwconv_fwd_raw(x, None, br_w, br_b, ACT_NONE, bf) if (br_w is not None and not fuse_tail) else None
        inv_act = ACT_NONE if fuse_tail else act
        if fourier:
            wr, wi = op_ws
            Co, Ci = wr.shape
            if w_pre is not None and tuple(w_pre.shape) == (2 * Co, 2 * Ci) and w_pre.is_contiguous():
                w = w_pre
            else:
                w = torch.empty((2 * Co, 2 * Ci), device=x.device, dtype=torch.float32)
                check(_lib.lib().hno_cmix_compose(ptr(wr), ptr(wi), ptr(w), Co, Ci, stream_ptr()), 'hno_cmix_compose')
            if Co == Ci == x.shape[1] and fourier_chain_supported(x, modes):
                s0, y = fourier_chain_fwd_raw(x, w, modes, 1.0 / n3, x2, inv_act)
            else:
                s0 = rfft3_crop_raw(x, modes, 1.0 / n3, False)
                s1 = pwconv_fwd_raw(s0, None, w, None, ACT_NONE)
                y = irfft3_pad_raw(s1, spatial, 1.0, True, x2, inv_act, ld=ld)
        else:
            (w,) = op_ws
            if spectral_chain_supported(x, modes, 1) and tuple(w.shape) == (x.shape[1], x.shape[1]):
                # the fused middle with one layer and no residual identity: z1 = selu(W z0) (hartley_operator.py:262-269)
                s0, zs, y = spectral_chain_fwd_raw(x, [w], modes, ACT_SELU, 1.0 / n3, inv_act, residual=0, addend=x2)
                s1 = zs[0]
            else:
                s0 = dht3_crop_raw(x, modes, 1.0 / n3)
                s1 = pwconv_fwd_raw(s0, None, w, None, ACT_SELU)      # SELU in the frequency domain (hartley_operator.py:262-269)
                y = pad_idht3_raw(s1, spatial, 1.0, x2, inv_act, ld=ld)
        if fuse_tail:
            sop, y = y, act_like(y)
            out = act_empty16(x.shape[0], 24, spatial, x.device, ld) if io16 else act_like(y)
            check(_lib.lib().hno_pwconv_fwd_branch(ptr(sop), ptr(x), ptr(br_w), ptr(br_b), ptr(cat_w), ptr(cat_b), ptr(y), ptr(out),
                                                   x.shape[0], 24, 24, 24, ld or _flat_v(x),
                                                   act | (ACT_BF16 if bf else 0) | (ACT_IO16 if io16 else 0), stream_ptr()), 'hno_pwconv_fwd_branch')
        else:
            out = pwconv_fwd_raw(y, x, cat_w, cat_b, act, bf)
        ctx.bf16 = bf
        ctx.io16 = io16
        ctx.save_for_backward(x, br_w, cat_w, w, s0, s1 if not fourier else None, y, out, br_b, cat_b, *op_ws)
        ctx.cfg = (bool(fourier), modes, act, spatial, n3, br_b is not None, cat_b is not None)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, br_w, cat_w, w, s0, s1, y, out, br_b, cat_b, *op_ws = ctx.saved_tensors
        fourier, modes, act, spatial, n3, br_has_b, cat_has_b = ctx.cfg
        late = ctx.leaf_params and _release_use(ctx, br_w, br_b, cat_w, cat_b, *op_ws) and _deferrable(br_w, br_b, cat_w, cat_b, *op_ws)
        d_br_w = d_br_b = None
        if br_w is not None and tuple(cat_w.shape[:2]) == (24, 48) and tuple(br_w.shape[:2]) == (24, 24):
            # one pass: p = d loss / d (s + x2) = g_y * act'(y); g_x = concat-path gradient + Wbr^T p; all four parameter gradients
            p, g_x, d_cat_w, d_cat_b, d_br_w, d_br_b = pwconv_bwd_branch_raw(g_out if ctx.io16 else _f32a(g_out), out, y, x, cat_w, br_w, act, act,
                                                                             defer=late, bf16=ctx.bf16, io16=ctx.io16)
            d_br_w = d_br_w.view_as(br_w)
            if not cat_has_b:
                d_cat_b = None
            if not br_has_b:
                d_br_b = None
        else:
            # p through the conv's xa_act product; the branch conv (its output gradient is p) adds its input gradient to g_x
            p, g_x, d_cat_w, d_cat_b = pwconv_bwd_raw(_f32a(g_out), out, y, x, cat_w, act, cat_has_b, xa_act=act, defer=late, bf16=ctx.bf16)
            if br_w is not None:
                _, _, d_br_w, d_br_b = pwconv_bwd_raw(p, None, x, None, br_w, ACT_NONE, br_has_b, accumulate_into=(g_x, None),
                                                      defer=late, bf16=ctx.bf16)
        if fourier:
            Co, Ci = w.shape[0] // 2, w.shape[1] // 2
            fused = Co == Ci == x.shape[1] and fourier_chain_supported(x, modes) and os.environ.get('HNO_FUSED_MID_BWD', '1') != '0'
            assert fused or not ctx.io16
            if fused:
                gx, dw2 = fourier_chain_bwd_raw(p, s0, w, modes, 1.0 / n3, g_x, defer=late, out16=ctx.io16)
            else:
                gs1 = rfft3_crop_raw(p, modes, 1.0, True)
                gs0, _, dw2, _ = pwconv_bwd_raw(gs1, None, s0, None, w, ACT_NONE, False)
            dwr = torch.empty((Co, Ci), device=x.device, dtype=torch.float32)
            dwi = torch.empty_like(dwr)
            # (round 5: with the fused middle the reduction of dW2 and this split join the batched end-of-backward launches)
            with _DeferReduce(late and fused) as d:
                check(_lib.lib().hno_cmix_split_grad_ex(ptr(dw2), ptr(dwr), ptr(dwi), Co, Ci, 1 if d.on else 0, stream_ptr()), 'hno_cmix_split_grad')
                d.keep(dw2)
            if not fused:
                gx = irfft3_pad_raw(gs0, spatial, 1.0 / n3, False, g_x, ACT_NONE, ld=chan_stride(x))
            d_ops = (dwr, dwi)
        else:
            if s1.dim() == 5 and spectral_chain_bwd_ok(x, modes, s0, s1.unsqueeze(0)):
                gx, dW = spectral_chain_bwd_raw(p, s0, op_ws, modes, ACT_SELU, 1.0 / n3, g_x, defer=late, residual=0)
                dw = dW[0]
            else:
                gs1 = dht3_crop_raw(p, modes, 1.0)
                gs0, _, dw, _ = pwconv_bwd_raw(gs1, s1, s0, None, w, ACT_SELU, False, defer=late)
                gx = pad_idht3_raw(gs0, spatial, 1.0 / n3, g_x, ACT_NONE, ld=chan_stride(x))
            d_ops = (dw,)
        return (gx, None, None, None, d_br_w, d_br_b, d_cat_w, d_cat_b) + d_ops + (None,) * (ctx.n_ops - len(d_ops))


class XSBlockFn(_HnoFunction):
    """One whole HNO-XS block (nets/hnosegxs.py:253-279) as a single autograd node:

        [mapping_conv(cat[x, skip])] -> TransformCrop -> n_XS x (z <- act((W + I) z)) -> PadInverse -> act
                                     -> conv_concat(cat[., block input])

    Owning the whole block lets the backward fuse the two gradients that meet at the block input
    (through the transform and through the concat skip) into the inverse-transform store instead
    of materialising both and adding them."""

    @staticmethod
    def meta(x, skip, map_w, map_b, cat_w, cat_b, modes, act, passthrough, nmap_w, nmap_b, nskip, *mix_ws):
        out = _m((x.shape[0], cat_w.shape[0]) + tuple(x.shape[2:]))
        return (out, x) if passthrough else out

    @staticmethod
    def forward(ctx, x, skip, map_w, map_b, cat_w, cat_b, modes, act, passthrough, nmap_w, nmap_b, nskip, *mix_ws):
        """passthrough: also return the block input `x` as a second output.  A later block that takes this
        tensor as its U-Net skip then sends its gradient HERE instead of to a second consumer edge of `x`, and
        the backward below folds it into the store of the concat-path gradient -- autograd's separate
        accumulation kernel (3 x 158 MB of traffic per step in HNOSeg-XS) disappears."""
        # nmap_w / nmap_b / nskip (round 4): the mapping_conv of the NEXT (decoder) block and its U-Net skip tensor.  The block then
        # returns xn = act(Wm [out ; nskip] + bm) instead of its own output: conv_concat and the next mapping_conv run as ONE pass
        # (hno_pwconv_fwd_chain / hno_pwconv_bwd_chain) and the next block is called without a mapping convolution.
        ctx.leaf_params = _leaf_params(ctx, map_w, map_b, cat_w, cat_b, nmap_w, nmap_b, *mix_ws)
        map_w, map_b, cat_w, cat_b, nmap_w, nmap_b = (_f32c(t) for t in (map_w, map_b, cat_w, cat_b, nmap_w, nmap_b))
        mix_ws = [_f32c(w) for w in mix_ws]
        spatial = tuple(x.shape[2:])
        if len(modes) == 2:                      # 2-D model on a (B, C, 1, H, W) view
            modes = (0,) + tuple(modes)
        modes = clamp_modes(modes, spatial)
        # a channel-padded input stays padded through the block when both transform directions take the stride
        x = _f32a(x) if (chan_stride(x) is not None and padded_ok(spatial, modes)) else _f32c(x)
        skip = to_layout(skip, chan_stride(x))
        _need_gpu(x, skip, cat_w, *mix_ws)
        has_map = map_w is not None
        bf = _autocast_bf16()       # autocast: bf16 operands for the two spatial convolutions of the block
        ctx.bf16 = bf
        xm = pwconv_fwd_raw(x, skip, map_w, map_b, act, bf) if has_map else x
        n3 = float(np.prod(spatial))
        if spectral_chain_supported(xm, modes, len(mix_ws)):
            z0, zs, u = spectral_chain_fwd_raw(xm, mix_ws, modes, act, 1.0 / n3, act)
        else:
            z0 = dht3_crop_raw(xm, modes, 1.0 / n3)
            zs = specmix_fwd_raw(z0, mix_ws, 1, act)
            u = pad_idht3_raw(zs[-1], spatial, 1.0, None, act, ld=chan_stride(xm))
        chained = nmap_w is not None
        xn = None
        if chained:
            # nskip given: the next block's mapping_conv (24 + 24 -> 24, same activation); nskip None: the model's conv_out behind the
            # last block (24 -> out_channels, no bias, no activation): xn are the low-resolution logits
            assert not passthrough and not bf
            C2 = int(nmap_w.shape[0])
            act2 = act if nskip is not None else ACT_NONE
            if nskip is not None:
                nskip = to_layout(_f32a(nskip), chan_stride(xm))
            # the concat convolution's output is read by the backward only: not stored when no input wants a gradient (inference)
            out = act_like(xm) if backward_wanted(ctx) else None
            xn = act_like(xm) if nskip is not None else act_empty(x.shape[0], C2, spatial, x.device, chan_stride(xm))
            check(_lib.lib().hno_pwconv_fwd_chain(ptr(u), ptr(xm), ptr(nskip), ptr(cat_w), ptr(cat_b), ptr(nmap_w), ptr(nmap_b), ptr(out), ptr(xn),
                                                  x.shape[0], int(cat_w.shape[0]), C2, chan_stride(xm) or _flat_v(xm), act, act2, stream_ptr()),
                  'hno_pwconv_fwd_chain')
            ctx.act2 = act2
        else:
            out = pwconv_fwd_raw(u, xm, cat_w, cat_b, act, bf)
        ctx.chain = chained
        # (the chained tensors go through save_for_backward like the others: an output kept on ctx directly would be a reference cycle)
        ctx.save_for_backward(x, skip, map_w, xm if has_map else None, z0, zs, u, cat_w, out, map_b, cat_b,
                              nmap_w, nmap_b, nskip if chained else None, xn, *mix_ws)
        ctx.cfg = (has_map, modes, act, spatial, n3, map_b is not None, cat_b is not None, bool(passthrough))
        ctx.set_materialize_grads(False)
        if chained:
            return xn
        if passthrough:
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, g_out, g_pass=None):
        x, skip, map_w, xm, z0, zs, u, cat_w, out, map_b, cat_b, nmap_w, nmap_b, nskip, xn, *mix_ws = ctx.saved_tensors
        has_map, modes, act, spatial, n3, map_has_b, cat_has_b, passthrough = ctx.cfg
        if not has_map:
            xm = x
        # weight gradients of leaves with .grad None are not read before backward ends: their slab reductions are batched
        lp = ctx.leaf_params and _release_use(ctx, map_w, map_b, cat_w, cat_b, nmap_w, nmap_b, *mix_ws)
        late_cat, late_mix, late_map = lp and _deferrable(cat_w, cat_b), lp and _deferrable(*mix_ws), lp and _deferrable(map_w, map_b)
        if g_out is None:
            raise _lib.HnoError('XSBlockFn.backward: no gradient for the block output')
        d_nmap_w = d_nmap_b = g_nskip = None
        chain_fused = False
        if ctx.chain:
            # g_out is the gradient of xn = act(Wm [out ; nskip] + bm).  One pass through both pointwise layers (hno_pwconv_bwd_chain:
            # the gradient between them never reaches memory); HNO_PW_CHAIN_BWD=0: the two layers apart (A/B)
            late_nmap = lp and _deferrable(nmap_w, nmap_b)
            ld_n = chan_stride(xm)
            C = int(cat_w.shape[0])
            if os.environ.get('HNO_PW_CHAIN_BWD', '1') != '0' and g_pass is None and chan_stride(u) == ld_n == chan_stride(xn):
                L = _lib.lib()
                gn = to_layout(g_out, ld_n)
                C2, has_k = int(nmap_w.shape[0]), nskip is not None
                cin2 = 2 * C if has_k else C
                g_u, g_skipin = act_like(u), act_like(xm)
                g_nskip = act_like(nskip) if has_k else None
                # [dWc | dbc | dWm | dbm]: the four parameters' order in the model -- under a data-parallel replica the kernel's output IS
                # their run of the flat gradient buffer (conv_out has no bias: its form ends behind dWm)
                nc = C * 2 * C + C
                nm = C2 * cin2 + (C2 if has_k else 0)
                plist = [cat_w, cat_b, nmap_w] + ([nmap_b] if has_k else [])
                if cat_has_b and all(t is not None for t in plist):
                    flat = _grad_buffer_concat(plist)
                else:
                    flat = torch.empty(nc + nm, device=u.device, dtype=torch.float32)
                ws = torch.empty(L.hno_pwconv_bwd_chain_workspace_bytes(C) // 4, device=u.device, dtype=torch.float32)
                with _DeferReduce(late_nmap and late_cat) as d:
                    check(L.hno_pwconv_bwd_chain(ptr(gn), ptr(xn) if ctx.act2 != ACT_NONE else None, ptr(out), ptr(nskip), ptr(u), ptr(xm), ptr(nmap_w),
                                                 ptr(cat_w), ptr(g_u), ptr(g_skipin), ptr(g_nskip), ptr(flat), ptr(ws), x.shape[0], C, C2,
                                                 ld_n or _flat_v(xm), act, ctx.act2, act | d.bit, stream_ptr()), 'hno_pwconv_bwd_chain')
                    d.keep(ws)
                d_cat_w, d_cat_b = flat[:C * 2 * C].view_as(cat_w), (flat[C * 2 * C:nc] if cat_has_b else None)
                d_nmap_w = flat[nc:nc + C2 * cin2].view_as(nmap_w)
                d_nmap_b = flat[nc + C2 * cin2:nc + C2 * cin2 + C2] if (nmap_b is not None and has_k) else None
                chain_fused = True
            else:
                g_out, g_nskip, d_nmap_w, d_nmap_b = pwconv_bwd_raw(to_layout(g_out, ld_n), xn if ctx.act2 != ACT_NONE else None, out, nskip,
                                                                   nmap_w.reshape(nmap_w.shape[0], -1), ctx.act2, nmap_b is not None,
                                                                   True, nskip is not None, defer=late_nmap, bias=nmap_b, bf16=False)
                d_nmap_w = d_nmap_w.view_as(nmap_w)
            if g_nskip is not None:
                g_nskip._hno_private = True      # fresh buffer of ours: the encoder block that owns the skip may accumulate into it
        # The passthrough gradient is accumulated IN PLACE into the buffer autograd handed us only when that buffer is
        # provably private: produced by our own mapping-conv backward below (tagged), i.e. not summed by the engine, not
        # captured by a hook, not shared with another consumer.  Anything else takes the out-of-place add.
        ld = chan_stride(xm)        # the block's activation layout (saved tensors keep their strides); gradients follow it
        private = g_pass is not None and getattr(g_pass, '_hno_private', False) and g_pass.dtype == torch.float32 \
            and (g_pass.is_contiguous() if ld is None else chan_stride(g_pass) == ld)
        g_pass = to_layout(g_pass, ld)
        _stats['pass_fused' if private else 'pass_unfused'] += g_pass is not None

        def plus_pass(t):
            out = act_like(t)
            check(_lib.lib().hno_add(ptr(t), ptr(g_pass), ptr(out), _ext(t), stream_ptr()), 'hno_add')
            return out
        # conv_concat backward; the SELU backward of PadInverse (g_u * act'(u)) is applied in its epilogue.  Without
        # a mapping conv the block input IS xm: the passthrough gradient is accumulated into the concat-path
        # gradient by the same kernel (gxb += ...).
        fuse_pass = private and not has_map
        if not chain_fused:
            g_u, g_skipin, d_cat_w, d_cat_b = pwconv_bwd_raw(to_layout(g_out, ld), out, u, xm, cat_w, act, cat_has_b, xa_act=act,
                                                             accumulate_into=(None, g_pass) if fuse_pass else None, defer=late_cat, bias=cat_b, bf16=ctx.bf16)
        if g_pass is not None and not has_map and not fuse_pass:
            g_skipin = plus_pass(g_skipin)
        if spectral_chain_bwd_ok(xm, modes, z0, zs):     # PadInverse^T, the layers' backward and TransformCrop^T + skip gradient
            g_xm, d_mix = spectral_chain_bwd_raw(g_u, z0, mix_ws, modes, act, 1.0 / n3, g_skipin, defer=late_mix)
        else:
            g_zl = dht3_crop_raw(g_u, modes, 1.0)                               # PadInverse^T
            g_z0, d_mix = specmix_bwd_raw(g_zl, z0, zs, mix_ws, 1, act, defer=late_mix)
            g_xm = pad_idht3_raw(g_z0, spatial, 1.0 / n3, g_skipin, ACT_NONE, ld=ld)    # TransformCrop^T + skip gradient
        d_mix = tuple(d_mix.unbind(0))
        if not has_map:
            return (g_xm, None, None, None, d_cat_w, d_cat_b, None, None, None, d_nmap_w, d_nmap_b, g_nskip) + d_mix
        g_x, g_skip, d_map_w, d_map_b = pwconv_bwd_raw(g_xm, xm, x, skip, map_w, act, map_has_b,
                                                       ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                                       accumulate_into=(g_pass, None) if private else None,
                                                       defer=late_map, bias=map_b, bf16=ctx.bf16)
        if g_pass is not None and not private and g_x is not None:
            g_x = plus_pass(g_x)
        if g_skip is not None:
            g_skip._hno_private = True          # fresh buffer of ours: a later consumer may accumulate into it
        return (g_x, g_skip, d_map_w, d_map_b, d_cat_w, d_cat_b, None, None, None, d_nmap_w, d_nmap_b, g_nskip) + d_mix


class ConvK2S2Fn(_HnoFunction):
    """Conv3d(k=2, s=2, p=1) + bias + act (conv_in; nets/hnosegxs.py:102-104,151)."""

    @staticmethod
    def meta(x, W, bias, act):
        return _m((x.shape[0], W.shape[0]) + tuple(v // 2 + 1 for v in x.shape[2:]))

    @staticmethod
    def forward(ctx, x, W, bias, act):
        x, W, bias = _f32c(x), _f32c(W), _f32c(bias)
        _need_gpu(x, W, bias)
        B, Cin, D, H, Wd = x.shape
        Cout = W.shape[0]
        so = (D // 2 + 1, H // 2 + 1, Wd // 2 + 1)
        # first producer of the HNOSeg-XS activations: channel-padded when the model asked for it (ops.channel_padded)
        ld = _pad_ld(np.prod(so)) if (getattr(_PAD, 'on', False) and so[0] > 1 and int(np.prod(so)) % 32) else None
        y = act_empty(B, Cout, so, x.device, ld)
        check(_lib.lib().hno_conv_k2s2_fwd(ptr(x), ptr(W), ptr(bias), ptr(y), B, Cin, Cout, D, H, Wd, act, ld or 0, stream_ptr()),
              'hno_conv_k2s2_fwd')
        ctx.save_for_backward(x, W, y, bias)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W, y, bias = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.HnoError('conv_in input gradient is not implemented (the image needs none)')
        ld = chan_stride(y)
        gy = to_layout(gy, ld)
        B, Cin, D, H, Wd = x.shape
        Cout = W.shape[0]
        dW = _grad_buffer(W)
        db = (_grad_buffer(bias) if bias is not None else torch.empty(Cout, device=W.device, dtype=torch.float32)) if ctx.has_bias else None
        ws = _wgrad_ws(Cin * 8, Cout, x.device)
        check(_lib.lib().hno_conv_k2s2_bwd(ptr(gy), ptr(y), ptr(x), ptr(W), None, ptr(dW), ptr(db), ptr(ws), B, Cin, Cout,
                                           D, H, Wd, ctx.act, ld or 0, stream_ptr()), 'hno_conv_k2s2_bwd')
        return None, dW, db, None


class StemChainFn(_HnoFunction):
    """conv_in = Conv3d(k 2, s 2, p 1) + bias + act and conv1 = Conv3d(k 1) + bias + act of HNOSeg-XS (nets/hnosegxs.py:102-108, 151-152) as
    one pass each way (hno_conv_k2s2_chain_fwd / _bwd): conv_in's output is neither written nor saved -- the backward recomputes its tile
    from the image it reads for conv_in's weight gradient anyway."""

    @staticmethod
    def meta(x, W, bias, W1, bias1, act):
        return _m((x.shape[0], W1.shape[0]) + tuple(v // 2 + 1 for v in x.shape[2:]))

    @staticmethod
    def supported(x, W, W1):
        return (x.ndim == 5 and x.is_cuda and tuple(W.shape[2:]) == (2, 2, 2) and W1.shape[1] == W.shape[0]
                and all(v == 1 for v in W1.shape[2:])
                and bool(_lib.lib().hno_conv_k2s2_chain_supported(int(W.shape[1]), int(W.shape[0]), int(W1.shape[0]))))

    @staticmethod
    def forward(ctx, x, W, bias, W1, bias1, act):
        ctx.leaf_params = _leaf_params(ctx, W, bias, W1, bias1)
        x, W, bias, W1, bias1 = _f32c(x), _f32c(W), _f32c(bias), _f32c(W1), _f32c(bias1)
        _need_gpu(x, W, W1)
        B, Cin, D, H, Wd = x.shape
        C0, C1 = int(W.shape[0]), int(W1.shape[0])
        so = (D // 2 + 1, H // 2 + 1, Wd // 2 + 1)
        ld = _pad_ld(np.prod(so)) if (getattr(_PAD, 'on', False) and so[0] > 1 and int(np.prod(so)) % 32) else None
        y1 = act_empty(B, C1, so, x.device, ld)
        check(_lib.lib().hno_conv_k2s2_chain_fwd(ptr(x), ptr(W), ptr(bias), ptr(W1), ptr(bias1), ptr(y1), B, Cin, C0, C1, D, H, Wd, act, act,
                                                 ld or 0, stream_ptr()), 'hno_conv_k2s2_chain_fwd')
        ctx.save_for_backward(x, W, bias, W1, bias1, y1)
        ctx.act = act
        return y1

    @staticmethod
    def backward(ctx, gy):
        x, W, bias, W1, bias1, y1 = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.HnoError('conv_in input gradient is not implemented (the image needs none)')
        lp = ctx.leaf_params and _release_use(ctx, W, bias, W1, bias1)
        late = lp and _deferrable(W, bias, W1, bias1)
        ld = chan_stride(y1)
        gy = to_layout(gy, ld)
        B, Cin, D, H, Wd = x.shape
        C0, C1 = int(W.shape[0]), int(W1.shape[0])
        L = _lib.lib()
        n_in, n_1 = C0 * Cin * 8, C1 * C0
        # [dW_in | db_in | dW1 | db1] is the order of the four parameters in the model: the kernel's output IS their run of the flat
        # gradient buffer when one is registered (data-parallel replicas)
        if bias is not None and bias1 is not None:
            flat = _grad_buffer_concat([W, bias, W1, bias1])
        else:
            flat = torch.empty(n_in + C0 + n_1 + C1, device=x.device, dtype=torch.float32)
        ws = torch.empty(L.hno_conv_k2s2_chain_bwd_workspace_bytes(Cin, C0, C1) // 4, device=x.device, dtype=torch.float32)
        with _DeferReduce(late) as d:
            check(L.hno_conv_k2s2_chain_bwd(ptr(gy), ptr(y1), ptr(x), ptr(W), ptr(bias), ptr(W1), ptr(flat), ptr(ws), B, Cin, C0, C1, D, H, Wd,
                                            ctx.act, ctx.act | d.bit, ld or 0, stream_ptr()), 'hno_conv_k2s2_chain_bwd')
            d.keep(ws)
        dW = flat[:n_in].view_as(W)
        db = flat[n_in:n_in + C0] if bias is not None else None
        dW1 = flat[n_in + C0:n_in + C0 + n_1].view_as(W1)
        db1 = flat[n_in + C0 + n_1:] if bias1 is not None else None
        return None, dW, db, dW1, db1, None


class UpSoftmaxFn(_HnoFunction):
    """trilinear(align_corners=False) upsample of K logits + softmax over channels
    (nets/hnosegxs.py:174-180 with conv_out commuted to low resolution)."""

    @staticmethod
    def meta(logits_lr, size, softmax):
        return _m(tuple(logits_lr.shape[:2]) + tuple(size))

    @staticmethod
    def forward(ctx, logits_lr, size, softmax):
        lr = _f32a(logits_lr)         # channel-padded logits are read in place (no repack kernel): the stride goes to the kernel
        _need_gpu(lr)
        B, K, d, h, w = lr.shape
        D, H, W = (int(s) for s in size)
        probs = torch.empty((B, K, D, H, W), device=lr.device, dtype=torch.float32)
        ld = chan_stride(lr)
        check(_lib.lib().hno_upsoftmax_fwd_ld(ptr(lr), ptr(probs), B, K, d, h, w, D, H, W, int(softmax), ld or 0, stream_ptr()),
              'hno_upsoftmax_fwd')
        ctx.lr_shape, ctx.softmax, ctx.lr_ld = tuple(lr.shape), int(softmax), ld
        if softmax:
            ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, g):
        probs = ctx.saved_tensors[0] if ctx.softmax else None
        g = _f32c(g)
        B, K, d, h, w = ctx.lr_shape
        D, H, W = g.shape[2:]
        nws = _lib.lib().hno_upsoftmax_bwd_workspace_bytes(B, K, d, h, w, D, H, W)
        ws = torch.empty(nws // 4, device=g.device, dtype=torch.float32) if nws else None
        ld = ctx.lr_ld if nws else None      # the gradient in the logits' own layout (padding zeroed by the kernel): separable form only
        g_lr = act_empty(B, K, (d, h, w), g.device, ld)
        check(_lib.lib().hno_upsoftmax_bwd_ld(ptr(g), ptr(probs), ptr(g_lr), ptr(ws), B, K, d, h, w, D, H, W, ctx.softmax, ld or 0,
                                              stream_ptr()), 'hno_upsoftmax_bwd')
        return g_lr, None, None


# ------------------------------------------------------------------------- inference head
_HEAD_MODE = threading.local()


@contextlib.contextmanager
def label_output():
    """Inside this context every model's output head returns uint8 class labels (B, 1, D, H, W) -- argmax over channels
    of the upsampled logits, fused into the upsampling kernel -- instead of probabilities.  Used by
    experiments.train_test.testing(); forward only (no autograd)."""
    prev = getattr(_HEAD_MODE, 'labels', False)
    _HEAD_MODE.labels = True
    try:
        yield
    finally:
        _HEAD_MODE.labels = prev


def up_argmax(logits_lr, size):
    lr = _f32a(logits_lr.detach())
    if lr.is_meta:
        return _m((lr.shape[0], 1) + tuple(size), torch.uint8)
    _need_gpu(lr)
    B, K, d, h, w = lr.shape
    D, H, W = (int(s) for s in size)
    labels = torch.empty((B, 1, D, H, W), device=lr.device, dtype=torch.uint8)
    check(_lib.lib().hno_up_argmax_ld(ptr(lr), ptr(labels), B, K, d, h, w, D, H, W, chan_stride(lr) or 0, stream_ptr()), 'hno_up_argmax')
    return labels


def output_act(output_activation):
    """(softmax?, elementwise activation id) of a model's ``output_activation`` argument: 'softmax' (fused into the
    upsampling kernel), None, or 'sigmoid' (multi-label outputs; an elementwise pass after the upsampling)."""
    name = output_activation if isinstance(output_activation, str) or output_activation is None \
        else getattr(output_activation, '__name__', None)
    if name == 'softmax':
        return True, ACT_NONE
    if name is None and output_activation is None:
        return False, ACT_NONE
    if name == 'sigmoid':
        return False, ACT_SIGMOID
    raise NotImplementedError(f'output activation {output_activation!r} is not provided by the HIP path (softmax, sigmoid or None)')


@contextlib.contextmanager
def expected_loss(labels_u8, loss_fn):
    """``with ops.expected_loss(labels_u8, loss_fn): y = model(x)`` followed by ``loss_fn(y, labels_u8)`` -- the reference's step
    (experiments/train_test.py:154-160) with one hint: the output head learns which loss is about to be applied to its probabilities and
    takes the loss sums while they are in registers (hno_uphead_loss_fwd), and the backward evaluates d loss / d probs inside the head's
    backward (hno_upsoftmax_loss_bwd) instead of writing and re-reading it.  The hint changes no result and no protocol: the head hands
    the finished loss to ``nets.custom_losses`` on the probabilities tensor, which uses it only for the same labels and loss; any other
    use of the probabilities takes the separate kernels as before.  Losses without ``hno_loss_spec`` (or ``HNO_HEAD_LOSS=0``) ignore it."""
    spec = getattr(loss_fn, 'hno_loss_spec', None)
    prev = getattr(_HEAD_MODE, 'loss', None)
    ok = (spec is not None and torch.is_tensor(labels_u8) and labels_u8.dtype == torch.uint8 and labels_u8.is_cuda
          and labels_u8.is_contiguous() and os.environ.get('HNO_HEAD_LOSS', '1') != '0')
    _HEAD_MODE.loss = (labels_u8, LOSS_KINDS[spec[0]], float(spec[1])) if ok else None
    try:
        yield
    finally:
        _HEAD_MODE.loss = prev


def precomputed_loss(y_pred, labels_u8, kind, param):
    """(loss, coef) the head left on its output for exactly this (labels, loss), else None"""
    hit = getattr(y_pred, '_hno_loss', None)
    if hit is None:
        return None
    lab, k, p, loss, coef = hit
    if k == kind and p == float(param) and lab.data_ptr() == labels_u8.data_ptr() and lab.numel() == labels_u8.numel():
        return loss, coef
    return None


def head_output(logits_lr, size, softmax, out_act=ACT_NONE):
    """Output head shared by all model families: probabilities (training / evaluation) or labels (label_output())."""
    if getattr(_HEAD_MODE, 'labels', False):
        return up_argmax(logits_lr, size)      # arg max of the logits == arg max after softmax / sigmoid
    hint = getattr(_HEAD_MODE, 'loss', None)
    if hint is not None and softmax and out_act == ACT_NONE and logits_lr.is_cuda and logits_lr.ndim == 5:
        lab, kind, param = hint
        B, K, d, h, w = logits_lr.shape
        D, H, W = (int(v) for v in size)
        if lab.numel() == B * D * H * W and lab.data_ptr() % 4 == 0 and _lib.lib().hno_uphead_loss_supported(B, K, d, h, w, D, H, W):
            y, loss, coef = HeadLossFn.apply(logits_lr, lab, size, kind, param)
            y._hno_loss = (lab, kind, param, loss, coef)
            return y
    y = UpSoftmaxFn.apply(logits_lr, size, softmax)
    return ActFn.apply(y, out_act) if out_act != ACT_NONE else y


class HeadLossFn(_HnoFunction):
    """Softmax head + PCC / Dice / ExpDice loss as one autograd node with outputs (probs, loss, coef): forward hno_uphead_loss_fwd,
    backward hno_upsoftmax_loss_bwd when only the loss sends a gradient (the training step); a gradient arriving at the probabilities
    as well is added to the loss's and takes the separate kernels."""

    @staticmethod
    def meta(logits_lr, labels_u8, size, kind, param):
        return _m(tuple(logits_lr.shape[:2]) + tuple(size)), _m(()), _m(tuple(logits_lr.shape[:2]) + (4,))

    @staticmethod
    def forward(ctx, logits_lr, labels_u8, size, kind, param):
        lr = _f32a(logits_lr)
        _need_gpu(lr, labels_u8)
        B, K, d, h, w = lr.shape
        D, H, W = (int(s) for s in size)
        L = _lib.lib()
        probs = torch.empty((B, K, D, H, W), device=lr.device, dtype=torch.float32)
        nws = L.hno_uphead_loss_workspace_doubles(B, K)
        stats = torch.empty(nws, device=lr.device, dtype=torch.float64)
        coef = torch.empty((B, K, 4), device=lr.device, dtype=torch.float32)
        loss = torch.empty((), device=lr.device, dtype=torch.float32)
        ld = chan_stride(lr)
        check(L.hno_uphead_loss_fwd(ptr(lr), ptr(labels_u8), ptr(probs), ptr(stats), nws, ptr(coef), ptr(loss), B, K, d, h, w, D, H, W,
                                    ld or 0, kind, float(param), stream_ptr()), 'hno_uphead_loss_fwd')
        ctx.lr_shape, ctx.lr_ld = tuple(lr.shape), ld
        ctx.save_for_backward(probs, labels_u8, coef)
        ctx.mark_non_differentiable(coef)
        ctx.set_materialize_grads(False)
        return probs, loss, coef

    @staticmethod
    def backward(ctx, g_probs, g_loss, _gcoef):
        if g_probs is None and g_loss is None:
            return None, None, None, None, None
        probs, labels_u8, coef = ctx.saved_tensors
        B, K, d, h, w = ctx.lr_shape
        D, H, W = probs.shape[2:]
        L = _lib.lib()
        nws = L.hno_upsoftmax_bwd_workspace_bytes(B, K, d, h, w, D, H, W)
        ws = torch.empty(nws // 4, device=probs.device, dtype=torch.float32) if nws else None
        ld = ctx.lr_ld if nws else None
        g_lr = act_empty(B, K, (d, h, w), probs.device, ld)
        if g_loss is not None:
            g_loss = _f32c(g_loss).reshape(1)
        if g_probs is None and nws and labels_u8.data_ptr() % 4 == 0:
            check(L.hno_upsoftmax_loss_bwd(ptr(probs), ptr(labels_u8), ptr(coef), ptr(g_loss), ptr(g_lr), ptr(ws), B, K, d, h, w, D, H, W,
                                           ld or 0, stream_ptr()), 'hno_upsoftmax_loss_bwd')
            return g_lr, None, None, None, None
        g = _f32c(g_probs) if g_probs is not None else None
        if g_loss is not None:
            gl = torch.empty_like(probs)
            check(L.hno_loss_bwd(ptr(probs), ptr(labels_u8), ptr(coef), ptr(g_loss), ptr(gl), B, K, _flat_v(probs), stream_ptr()), 'hno_loss_bwd')
            g = gl if g is None else g + gl
        check(L.hno_upsoftmax_bwd_ld(ptr(g), ptr(probs), ptr(g_lr), ptr(ws), B, K, d, h, w, D, H, W, 1, ld or 0, stream_ptr()),
              'hno_upsoftmax_bwd')
        return g_lr, None, None, None, None


class SegLossFn(_HnoFunction):
    """PCC / Dice / ExpDice on uint8 labels (nets/custom_losses.py:17-133 with the one-hot
    encoding of experiments/utils.py:74-97 fused)."""

    @staticmethod
    def meta(probs, labels_u8, kind, param):
        return _m(()), _m(tuple(probs.shape[:2]) + (4,))

    @staticmethod
    def forward(ctx, probs, labels_u8, kind, param):
        probs = _f32c(probs)
        _need_gpu(probs, labels_u8)
        assert labels_u8.dtype == torch.uint8 and labels_u8.is_contiguous()
        B, K = probs.shape[:2]
        V = _flat_v(probs)
        assert labels_u8.numel() == B * V
        nws = _lib.lib().hno_loss_workspace_doubles(B, K, V)
        stats = torch.empty(nws, device=probs.device, dtype=torch.float64)
        coef = torch.empty((B, K, 4), device=probs.device, dtype=torch.float32)
        loss = torch.empty((), device=probs.device, dtype=torch.float32)
        check(_lib.lib().hno_loss_fwd_ws(ptr(probs), ptr(labels_u8), ptr(stats), nws, ptr(coef), ptr(loss), B, K, V, kind,
                                         float(param), stream_ptr()), 'hno_loss_fwd_ws')
        ctx.save_for_backward(probs, labels_u8, coef)
        ctx.mark_non_differentiable(coef)
        ctx.set_materialize_grads(False)      # no zero tensor (a fill kernel per step) for the coefficient output's gradient
        return loss, coef

    @staticmethod
    def backward(ctx, gl, _gcoef):
        if gl is None:
            return None, None, None, None
        probs, labels_u8, coef = ctx.saved_tensors
        B, K = probs.shape[:2]
        V = _flat_v(probs)
        gl = _f32c(gl).reshape(1)
        g = torch.empty_like(probs)
        check(_lib.lib().hno_loss_bwd(ptr(probs), ptr(labels_u8), ptr(coef), ptr(gl), ptr(g), B, K, V, stream_ptr()),
              'hno_loss_bwd')
        return g, None, None, None


# ------------------------------------------------------------------------- label helpers
_label_maps = {}


_ones = {}


def backward_from(loss):
    """``loss.backward()`` without the fill kernel autograd launches for its root gradient (``torch.ones_like(loss)`` every step): the
    same d loss / d loss = 1, from a tensor kept per (device, dtype)."""
    key = (loss.device, loss.dtype)
    one = _ones.get(key)
    if one is None:
        one = _ones[key] = torch.ones((), device=loss.device, dtype=loss.dtype)
    loss.backward(gradient=one.expand_as(loss) if loss.dim() else one)


def sum_pairs(entries):
    """entries: (dst, a, b, scale) of contiguous fp32 GPU tensors with equal element counts; dst = scale * (a + b), elementwise, for ALL
    entries in one launch per 64 tensors (hno_sum_pairs).  Capturable: the addresses travel in the kernel arguments.  Used for the join of
    the two half-batch passes of a captured training step (gradients: dst = a, scale 1; the loss: scale 0.5)."""
    import ctypes
    if not entries:
        return
    n = len(entries)
    for d, a, b, _ in entries:
        _need_gpu(d, a, b)
        assert d.dtype == a.dtype == b.dtype == torch.float32 and d.numel() == a.numel() == b.numel()
        assert d.is_contiguous() and a.is_contiguous() and b.is_contiguous()
    vp = ctypes.c_void_p * n
    check(_lib.lib().hno_sum_pairs(vp(*[e[0].data_ptr() for e in entries]), vp(*[e[1].data_ptr() for e in entries]),
                                   vp(*[e[2].data_ptr() for e in entries]), (ctypes.c_longlong * n)(*[e[0].numel() for e in entries]),
                                   (ctypes.c_float * n)(*[float(e[3]) for e in entries]), n, stream_ptr()), 'hno_sum_pairs')


def labels_prepare(labels, num_classes, mapping=None, want_onehot=False):
    """(B,1,...) float/int labels -> uint8 class map (B,...) [+ one-hot fp32 (B,K,...)] on the GPU
    (experiments/utils.py:74-119)."""
    assert labels.shape[1] == 1, 'Can only handle single label per pixel.'
    lab = _f32c(labels)
    if lab.is_meta:
        u8 = _m((lab.shape[0],) + tuple(lab.shape[2:]), torch.uint8)
        return (u8, _m((lab.shape[0], num_classes) + tuple(lab.shape[2:]))) if want_onehot else u8
    _need_gpu(lab)
    B, V = lab.shape[0], _flat_v(lab)
    u8 = torch.empty((B,) + tuple(lab.shape[2:]), device=lab.device, dtype=torch.uint8)
    onehot = torch.empty((B, num_classes) + tuple(lab.shape[2:]), device=lab.device, dtype=torch.float32) if want_onehot else None
    rf = rt = None
    n = 0
    if mapping:
        # the two small tables go to the device ONCE per (mapping, device): a host-to-device copy per step is a synchronising copy in the
        # reference's loop and is not allowed inside a captured step (experiments.train_test.CapturedStep)
        key = (tuple(mapping.items()), str(lab.device))
        hit = _label_maps.get(key)
        if hit is None:
            if len(_label_maps) > 16:
                _label_maps.clear()
            hit = _label_maps[key] = (torch.tensor(list(mapping.keys()), device=lab.device, dtype=torch.int32),
                                      torch.tensor(list(mapping.values()), device=lab.device, dtype=torch.int32))
        rf, rt = hit
        n = len(mapping)
    check(_lib.lib().hno_labels_prepare(ptr(lab), ptr(rf), ptr(rt), n, ptr(u8), ptr(onehot), B, num_classes, V, stream_ptr()),
          'hno_labels_prepare')
    return (u8, onehot) if want_onehot else u8


def onehot_to_u8(y_true):
    """One-hot (B,K,...) fp32 -> uint8 class map (drop-in path for reference-style callers)."""
    return torch.argmax(y_true, dim=1).to(torch.uint8).contiguous()
