"""bf16 matrix-core path (the reference's ``use_autocast`` mode, experiments/train_test.py:79,154-168) over the
``hno_cb_*`` entry points of include/hno.h.

Activations are channels-last bf16 tensors (B, D, H, W, C) -- torch tensors as containers only; parameters, their
gradients and the GroupNorm statistics are fp32.  Every Function here is one launch group of libhno; nothing
computes through ATen, and there is no CPU path (meta tensors get shape inference only, see ops._HnoFunction).
"""
import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .ops import _HnoFunction, _need_gpu, _m, ACT_NONE

BF16 = torch.bfloat16
import os as _os
_LAZY_STATS = _os.environ.get('HNO_LAZY_STATS', '1') != '0'     # A/B switch: GroupNorm statistics finished by gn_apply (round 4)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), device=device, dtype=torch.uint8)


def _cl(t):
    """contiguous channels-last bf16 container check"""
    assert t.dtype == BF16 and t.is_contiguous() and t.ndim == 5, 'expected a contiguous (B, D, H, W, C) bf16 tensor'
    return t


def _f32(t):
    if t is None:
        return None
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def autocast_bf16():
    """True inside ``torch.autocast('cuda', dtype=torch.bfloat16)`` -- how the reference's ``use_autocast`` reaches the
    model (train_test.py:154-160).  fp16 autocast (torch's CUDA default dtype) is refused: this path is bf16."""
    if not torch.is_autocast_enabled('cuda'):
        return False
    dt = torch.get_autocast_dtype('cuda')
    if dt != torch.bfloat16:
        raise NotImplementedError(f'autocast dtype {dt} is not provided by the HIP path: use torch.autocast("cuda", dtype=torch.bfloat16)')
    return True


# ----------------------------------------------------------------------------------------------- raw launchers
def pack_weights(W, role, Cin, Cout, ks):
    """fp32 parameter -> packed bf16 GEMM operand.  role 0 conv fwd, 1 conv dgrad, 2 ConvTranspose fwd, 3 ConvTranspose dgrad."""
    L = _lib.lib()
    W = _f32(W)
    out = _ws(L.hno_cb_packed_weight_bytes(Cin if role in (0, 2) else Cout, Cout if role in (0, 2) else Cin, ks), W.device)
    check(L.hno_cb_pack_weights(ptr(W), ptr(out), role, Cin, Cout, ks, stream_ptr()), 'hno_cb_pack_weights')
    return out


def pack_weights_both(W, transposed, Cin, Cout, ks):
    """forward operand and input-gradient operand of one layer in one launch"""
    L = _lib.lib()
    W = _f32(W)
    wf = _ws(L.hno_cb_packed_weight_bytes(Cin, Cout, ks), W.device)
    wb = _ws(L.hno_cb_packed_weight_bytes(Cout, Cin, ks), W.device)
    check(L.hno_cb_pack_weights_both(ptr(W), ptr(wf), ptr(wb), int(transposed), Cin, Cout, ks, stream_ptr()), 'hno_cb_pack_weights_both')
    return wf, wb


class PackedWeights:
    """Packed bf16 GEMM operands (forward + input gradient) of a set of conv layers, refreshed by ONE launch per step
    (hno_cb_pack_weights_multi) instead of one per layer: buffers and the device table are built once per (layer set, weight
    storage) and reused; `refresh()` re-reads the current fp32 weights.  ConvFn looks the buffers up by the weight's storage."""
    _current = None          # the instance whose buffers ConvFn may use (set by refresh(), valid for this forward + backward)

    def __init__(self, layers):
        import ctypes
        import weakref
        import numpy as np
        L = _lib.lib()
        self.entries = {}
        rows = []
        for op in layers:
            W = op.weight
            transposed = isinstance(op, torch.nn.ConvTranspose3d)
            Cin, Cout = (W.shape[0], W.shape[1]) if transposed else (W.shape[1], W.shape[0])
            ks = int(op.kernel_size[0])
            if Cin % 8 or Cout % 8 or W.dtype != torch.float32 or not W.is_contiguous():
                continue
            wf = _ws(L.hno_cb_packed_weight_bytes(Cin, Cout, ks), W.device).zero_()     # padding rows / channels stay zero:
            wb = _ws(L.hno_cb_packed_weight_bytes(Cout, Cin, ks), W.device).zero_()     # the refresh kernel only writes real weights
            row = np.zeros(16, dtype=np.int64)
            check(L.hno_cb_pack_table_row(row.ctypes.data_as(ctypes.c_void_p), ptr(W), ptr(wf), ptr(wb), int(transposed), Cin, Cout, ks),
                  'hno_cb_pack_table_row')
            rows.append(row)
            self.entries[W.data_ptr()] = (weakref.ref(W), wf, wb)
        self.key = tuple(self.entries)
        self.table = torch.from_numpy(np.stack(rows)).to(layers[0].weight.device) if rows else None
        self.chunks = int(sum(L.hno_cb_pack_row_chunks(r.ctypes.data_as(ctypes.c_void_p)) for r in rows))

    def refresh(self):
        self.generation = getattr(self, 'generation', 0) + 1    # ConvFn.backward re-packs when the buffers were rewritten since its forward
        if self.table is not None:
            check(_lib.lib().hno_cb_pack_weights_multi(ptr(self.table), self.table.shape[0], self.chunks, stream_ptr()),
                  'hno_cb_pack_weights_multi')
        PackedWeights._current = self

    @staticmethod
    def release():
        """end of the owning model's forward: no other caller may pick these buffers up (a later tensor can reuse a freed
        parameter's address)"""
        PackedWeights._current = None

    @staticmethod
    def lookup(W):
        cur = PackedWeights._current
        if cur is None:
            return None
        e = cur.entries.get(W.data_ptr())
        if e is None or e[0]() is not W:          # the very Parameter object the table was built from, not just its address
            return None
        return e[1], e[2]


def conv_raw(xa, xb, wpacked, bias, Cout, out_spatial, mode, ks, stride, pad, want_stats, eps=1e-5, lazy=False):
    """gather GEMM; -> (y (B, Do, Ho, Wo, Cout) bf16, mean_rstd (B, 2) fp32 or None[, nstat]).  lazy: the statistics are left as
    per-workgroup partials behind the (B, 2) slot (same storage) and their count is returned as a third value; gn_apply_raw(...,
    nstat=) finishes them -- one dependent launch per layer less."""
    _need_gpu(xa, xb)
    xa = _cl(xa)
    B, Di, Hi, Wi, Ca = xa.shape
    Cb = 0
    if xb is not None:
        xb = _cl(xb)
        assert xb.shape[:4] == xa.shape[:4]
        Cb = xb.shape[4]
    Do, Ho, Wo = (int(v) for v in out_spatial)
    L = _lib.lib()
    y = torch.empty((B, Do, Ho, Wo, Cout), device=xa.device, dtype=BF16)
    lazy = bool(lazy and want_stats)
    mr = nstat = None
    if lazy:
        import ctypes
        full = torch.empty(L.hno_cb_conv_stats_floats(B, Cout, Do, Ho, Wo), device=xa.device, dtype=torch.float32)
        mr = full[:2 * B].view(B, 2)            # (mean, rstd) slot, written by gn_apply; the partials follow in the same storage
        nstat = ctypes.c_int(0)
    elif want_stats:
        mr = torch.empty((B, 2), device=xa.device, dtype=torch.float32)
    nws = L.hno_cb_conv_workspace_bytes(B, Ca + Cb, Cout, Do, Ho, Wo, ks)
    ws = _ws(nws, xa.device)
    check(L.hno_cb_conv(ptr(xa), Ca, ptr(xb), Cb, ptr(wpacked), ptr(_f32(bias)), ptr(y), ptr(mr), float(eps), ptr(ws), nws, mode, B, Cout,
                        Di, Hi, Wi, Do, Ho, Wo, ks, stride, pad, nstat, stream_ptr()), 'hno_cb_conv')
    if lazy:
        return y, mr, int(nstat.value)
    return y, mr


def conv_split_raw(x, wpacked, Cout, c_split, out_spatial, mode, ks, stride, pad):
    """conv_raw without statistics whose output channels go to TWO contiguous tensors: [0, c_split) and [c_split, Cout)
    (hno_cb_conv_split): the input gradient of a two-input convolution, one tensor per input."""
    _need_gpu(x)
    x = _cl(x)
    B, Di, Hi, Wi, Ca = x.shape
    Do, Ho, Wo = (int(v) for v in out_spatial)
    L = _lib.lib()
    ya = torch.empty((B, Do, Ho, Wo, c_split), device=x.device, dtype=BF16)
    yb = torch.empty((B, Do, Ho, Wo, Cout - c_split), device=x.device, dtype=BF16)
    nws = L.hno_cb_conv_workspace_bytes(B, Ca, Cout, Do, Ho, Wo, ks)
    ws = _ws(nws, x.device)
    check(L.hno_cb_conv_split(ptr(x), Ca, None, 0, ptr(wpacked), None, ptr(ya), ptr(yb), int(c_split), ptr(ws), nws, mode, B, Cout,
                              Di, Hi, Wi, Do, Ho, Wo, ks, stride, pad, stream_ptr()), 'hno_cb_conv_split')
    return ya, yb


def wgrad_raw(g, xa, xb, w_shape, transposed, ks, stride, pad, param=None):
    """dW (fp32, parameter layout).  conv: g on the output grid, x = [xa ; xb] its input.  transposed: x the ConvTranspose input."""
    g, xa = _cl(g), _cl(xa)
    L = _lib.lib()
    Cb = 0 if xb is None else xb.shape[4]
    dW = ops._grad_buffer(param) if (param is not None and tuple(param.shape) == tuple(w_shape)) else \
        torch.empty(tuple(w_shape), device=g.device, dtype=torch.float32)
    Cin, Cout = (xa.shape[4] + Cb, g.shape[4])
    nws = L.hno_cb_wgrad_workspace_bytes(Cin, Cout, ks)
    ws = _ws(nws, g.device)
    check(L.hno_cb_wgrad(ptr(g), g.shape[4], ptr(xa), xa.shape[4], ptr(xb), Cb, ptr(dW), ptr(ws), nws, int(transposed), g.shape[0],
                         xa.shape[1], xa.shape[2], xa.shape[3], g.shape[1], g.shape[2], g.shape[3], ks, stride, pad, stream_ptr()),
          'hno_cb_wgrad')
    return dW


def colsum_raw(g):
    C = g.shape[-1]
    L = _lib.lib()
    out = torch.empty(C, device=g.device, dtype=torch.float32)
    ws = _ws(L.hno_cb_colsum_workspace_bytes(C), g.device)
    check(L.hno_cb_colsum(ptr(g), ptr(out), ptr(ws), C, g.numel() // C, stream_ptr()), 'hno_cb_colsum')
    return out


def gn_apply_raw(y1, mr1, g1, b1, act, y2=None, mr2=None, g2=None, b2=None, nstat1=0, nstat2=0, eps=1e-5):
    y1 = _cl(y1)
    B, C = y1.shape[0], y1.shape[4]
    V = y1.shape[1] * y1.shape[2] * y1.shape[3]
    z = torch.empty_like(y1)
    check(_lib.lib().hno_cb_gn_apply(ptr(y1), ptr(mr1), ptr(_f32(g1)), ptr(_f32(b1)), ptr(y2), ptr(mr2), ptr(_f32(g2)), ptr(_f32(b2)), ptr(z),
                                     B, C, V, act, int(nstat1), int(nstat2), float(eps), stream_ptr()), 'hno_cb_gn_apply')
    return z


_stats = {'colsum_fused': 0, 'colsum_pass': 0}      # how ConvFn.backward got its bias gradient (tests)


def gn_bwd_raw(dz, y, mr, gamma, beta, act, colsum=False):
    """colsum: also return sum_{b, v} dy[b, v, c] (the bias gradient of the convolution that produced y), which the kernels get from
    their per-channel reductions without another pass over dy."""
    dz, y = _cl(dz), _cl(y)
    B, C = y.shape[0], y.shape[4]
    V = y.shape[1] * y.shape[2] * y.shape[3]
    L = _lib.lib()
    dy = torch.empty_like(y)
    dg = torch.empty(C, device=y.device, dtype=torch.float32)
    db = torch.empty(C, device=y.device, dtype=torch.float32)
    cs = torch.empty(C, device=y.device, dtype=torch.float32) if colsum else None
    ws = _ws(L.hno_cb_gn_bwd_workspace_bytes(B, C), y.device)
    check(L.hno_cb_gn_bwd(ptr(dz), ptr(y), ptr(mr), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(dy), ptr(dg), ptr(db), ptr(cs), ptr(ws), B, C, V,
                          act, 0, stream_ptr()), 'hno_cb_gn_bwd')
    return (dy, dg, db, cs) if colsum else (dy, dg, db)


def pack_input_raw(x, CP=None):
    """fp32 (B, C, D, H, W) -> bf16 (B, D, H, W, CP), pad channels zero"""
    x = _f32(x)
    _need_gpu(x)
    B, C = x.shape[:2]
    CP = CP or (C + 7) // 8 * 8
    y = torch.empty((B,) + tuple(x.shape[2:]) + (CP,), device=x.device, dtype=BF16)
    check(_lib.lib().hno_cb_pack_input(ptr(x), ptr(y), B, C, CP, x[0, 0].numel(), stream_ptr()), 'hno_cb_pack_input')
    return y


def unpack_raw(x, C=None):
    """bf16 (B, D, H, W, CP) -> fp32 (B, C, D, H, W)"""
    x = _cl(x)
    B, CP = x.shape[0], x.shape[4]
    C = C or CP
    y = torch.empty((B, C) + tuple(x.shape[1:4]), device=x.device, dtype=torch.float32)
    check(_lib.lib().hno_cb_unpack(ptr(x), ptr(y), B, C, CP, x.shape[1] * x.shape[2] * x.shape[3], stream_ptr()), 'hno_cb_unpack')
    return y


# ----------------------------------------------------------------------------------------------- autograd
def _out_spatial(in_sp, ks, stride, transposed):
    if transposed:
        return tuple(2 * v for v in in_sp)                      # k 3, s 2, p 1, output_padding 1 (nets_utils.py:195-203)
    if ks == 1:
        return tuple(in_sp)
    if ks == 2:
        return tuple(v // 2 + 1 for v in in_sp)                 # k 2, s 2, p 1 (conv_in)
    return tuple((v - 1) // stride + 1 for v in in_sp)          # k 3, p 1


class ConvFn(_HnoFunction):
    """Conv3d / ConvTranspose3d of ConvNormAct / ConvTransposeNormAct (nets/nets_utils.py:136-211) on channels-last bf16 with
    the GroupNorm(1, C) statistics of the output produced in the same pass.

        forward(xa, xb, W, bias, ks, stride, transposed, want_stats, eps) -> (y, mean_rstd)

    xb: optional second input concatenated after xa along channels (the decoder's skip, architectures.py:236-240).
    backward: input gradient = the same gather GEMM with the channel roles swapped; weight gradient = hno_cb_wgrad;
    bias gradient = column sums."""

    @staticmethod
    def meta(xa, xb, W, bias, ks, stride, transposed, want_stats, eps, lazy=False):
        Cout = W.shape[1] if transposed else W.shape[0]
        osz = _out_spatial(tuple(xa.shape[1:4]), ks, stride, transposed)
        out = _m((xa.shape[0],) + osz + (Cout,), BF16), (_m((xa.shape[0], 2)) if want_stats else None)
        return out + (0,) if (lazy and want_stats) else out

    @staticmethod
    def forward(ctx, xa, xb, W, bias, ks, stride, transposed, want_stats, eps, lazy=False):
        """lazy (with want_stats): -> (y, mean_rstd, nstat): the statistics are unfinished partials (conv_raw) for GNActFn"""
        _need_gpu(xa, xb, W, bias)
        Ca = xa.shape[4]
        Cb = xb.shape[4] if xb is not None else 0
        Cin = Ca + Cb
        Cout = W.shape[1] if transposed else W.shape[0]
        pad = 0 if ks == 1 else 1
        osz = _out_spatial(tuple(xa.shape[1:4]), ks, stride, transposed)
        need_dgrad = ctx.needs_input_grad[0] or (xb is not None and ctx.needs_input_grad[1])
        pre = PackedWeights.lookup(W)
        ctx.wp_owner = None
        if pre is not None:   # packed with every other layer by this forward's PackedWeights.refresh()
            wp, ctx.wpd = pre
            ctx.wp_owner = (PackedWeights._current, PackedWeights._current.generation)
        elif need_dgrad:      # the weights do not change between forward and backward: pack both GEMM operands now, in one launch
            wp, ctx.wpd = pack_weights_both(W, transposed, Cin, Cout, ks)
        else:
            wp, ctx.wpd = pack_weights(W, 2 if transposed else 0, Cin, Cout, ks), None
        res = conv_raw(xa, xb, wp, bias, Cout, osz, 1 if transposed else 0, ks, stride, pad, want_stats, eps, lazy=lazy)
        y, mr = res[0], res[1]
        ctx.save_for_backward(xa, xb, W)
        ctx.cfg = (ks, stride, bool(transposed), pad, bias is not None, Ca, Cb, Cout)
        ctx.mark_non_differentiable(*([mr] if mr is not None else []))
        ctx.set_materialize_grads(False)      # no zero tensor (one fill kernel per layer) for the statistics output's gradient
        return res

    @staticmethod
    def backward(ctx, gy, _gmr=None, _gn=None):
        if gy is None:
            return (None,) * 10
        xa, xb, W = ctx.saved_tensors
        ks, stride, transposed, pad, has_bias, Ca, Cb, Cout = ctx.cfg
        # left by GNActFn.backward: the column sums of this very tensor -- valid only while nobody wrote into it (the autograd engine
        # may accumulate a second consumer's gradient IN PLACE into the first-arrived tensor: that bumps its version counter)
        tag = getattr(gy, '_hno_colsum', None)
        cs = tag[0] if (tag is not None and gy._version == tag[1]) else None
        gy = gy.contiguous()
        Cin = Ca + Cb
        gxa = gxb = None
        if ctx.needs_input_grad[0] or (xb is not None and ctx.needs_input_grad[1]):
            # the shared packed buffers are rewritten by every forward of the owning model: a backward that runs after ANOTHER forward
            # (two graphs alive, a validation forward in between, gradient accumulation) packs its own operand from the saved weights
            stale = ctx.wp_owner is not None and ctx.wp_owner[0].generation != ctx.wp_owner[1]
            wp = ctx.wpd if (ctx.wpd is not None and not stale) else pack_weights(W, 3 if transposed else 1, Cin, Cout, ks)
            if xb is None or _os.environ.get('HNO_CB_SPLIT', '1') == '0':
                gx, _ = conv_raw(gy, None, wp, None, Cin, tuple(xa.shape[1:4]), 0 if transposed else 1, ks, stride, pad, False)
                if xb is None:
                    gxa = gx
                else:   # (A/B: views of one tensor; the consumers copy)
                    gxa, gxb = gx[..., :Ca], gx[..., Ca:]
            else:   # round 5: the two inputs' gradients leave the GEMM as two contiguous tensors (the decoder's two-input convolutions)
                gxa, gxb = conv_split_raw(gy, wp, Cin, Ca, tuple(xa.shape[1:4]), 0 if transposed else 1, ks, stride, pad)
        dW = wgrad_raw(gy, xa, xb, W.shape, transposed, ks, stride, pad, param=W if W.is_leaf else None)
        db = None
        if has_bias:
            db = cs if (cs is not None and cs.numel() == Cout) else colsum_raw(gy)
            _stats['colsum_fused' if db is cs else 'colsum_pass'] += 1
        return gxa, gxb, dW, db, None, None, None, None, None, None


class GNActFn(_HnoFunction):
    """z = act(GroupNorm(1, C)(y1)) [+ act(GroupNorm(1, C)(y2))] on channels-last bf16 with the statistics from ConvFn
    (nets/nets_utils.py:127-133; the two-branch form is the residual sum of a V-Net section, architectures.py:205-224)."""

    @staticmethod
    def meta(y1, mr1, g1, b1, act, y2=None, mr2=None, g2=None, b2=None, nstat1=0, nstat2=0, eps=1e-5):
        return _m(y1.shape, BF16)

    @staticmethod
    def forward(ctx, y1, mr1, g1, b1, act, y2=None, mr2=None, g2=None, b2=None, nstat1=0, nstat2=0, eps=1e-5):
        """nstat > 0: mr holds ConvFn's lazy statistics (finished by the kernel, which also stores mean / rstd for the backward)"""
        _need_gpu(y1, y2)
        z = gn_apply_raw(y1, mr1, g1, b1, act, y2, mr2, g2, b2, nstat1, nstat2, eps)
        ctx.save_for_backward(y1, mr1, g1, b1, y2, mr2, g2, b2)
        ctx.act = act
        return z

    @staticmethod
    def backward(ctx, dz):
        y1, mr1, g1, b1, y2, mr2, g2, b2 = ctx.saved_tensors
        dz = dz.contiguous()
        # the column sums ride on the gradient tensor: the producing ConvFn's backward takes them as its bias gradient when autograd
        # hands it this very tensor (one consumer; a summed gradient is a new tensor without the attribute)
        dy1, dg1, db1, cs1 = gn_bwd_raw(dz, y1, mr1, g1, b1, ctx.act, colsum=True)
        dy1._hno_colsum = (cs1, dy1._version)
        dy2 = dg2 = db2 = None
        if y2 is not None:
            dy2, dg2, db2, cs2 = gn_bwd_raw(dz, y2, mr2, g2, b2, ctx.act, colsum=True)
            dy2._hno_colsum = (cs2, dy2._version)
        return dy1, None, dg1, db1, None, dy2, None, dg2, db2, None, None, None


class PackInputFn(_HnoFunction):
    """fp32 NCDHW -> bf16 channels-last (the autocast cast of the first layer's input)."""

    @staticmethod
    def meta(x, CP):
        return _m((x.shape[0],) + tuple(x.shape[2:]) + (CP,), BF16)

    @staticmethod
    def forward(ctx, x, CP):
        ctx.C = x.shape[1]
        return pack_input_raw(x, CP)

    @staticmethod
    def backward(ctx, g):
        return unpack_raw(g.contiguous(), ctx.C), None


class UnpackFn(_HnoFunction):
    """bf16 channels-last -> fp32 NCDHW (where the bf16 region hands over to the fp32 head)."""

    @staticmethod
    def meta(x, C):
        return _m((x.shape[0], C) + tuple(x.shape[1:4]))

    @staticmethod
    def forward(ctx, x, C):
        ctx.CP = x.shape[4]
        return unpack_raw(x, C)

    @staticmethod
    def backward(ctx, g):
        return pack_input_raw(g, ctx.CP), None


# ----------------------------------------------------------------------------------------------- layer helpers
def conv_norm_act(layer, xa, xb=None, residual=None):
    """ConvNormAct / ConvTransposeNormAct (nets/nets_utils.py:136-211) on channels-last bf16: conv (+ fused GroupNorm
    statistics) then GroupNorm + activation.  `residual` = (y2, mr2, norm2): a second pre-normalisation branch whose activated
    value is added (the V-Net section's residual sum).  -> activated tensor, or (y, mean_rstd) when `layer` is to be used as
    the residual branch of another call (pass residual='defer')."""
    op = layer.op
    transposed = isinstance(op, torch.nn.ConvTranspose3d)
    if op.weight.ndim != 5:
        raise NotImplementedError('the bf16 path is 3-D only')
    ks, stride = int(op.kernel_size[0]), int(op.stride[0])
    norm = layer.normalization
    act = ops.act_id(layer.activation)
    W, b = op.weight, op.bias
    Cin_w = W.shape[0] if transposed else W.shape[1]
    Cin_x = xa.shape[4] + (xb.shape[4] if xb is not None else 0)
    if Cin_w < Cin_x:          # first layer: image channels were padded to a multiple of 8 (zero channels, zero weights)
        assert not transposed and xb is None
        W = torch.nn.functional.pad(W, (0, 0, 0, 0, 0, 0, 0, Cin_x - Cin_w))
    eps = norm.eps if norm is not None else 1e-5
    nstat = 0
    if norm is not None and not xa.is_meta and _LAZY_STATS:      # lazy statistics: GNActFn's kernel finishes them (no finalize launch)
        y, mr, nstat = ConvFn.apply(xa, xb, W, b, ks, stride, transposed, True, eps, True)
    else:
        y, mr = ConvFn.apply(xa, xb, W, b, ks, stride, transposed, norm is not None, eps)
    if isinstance(residual, str):
        return y, mr, norm, nstat
    if norm is None:           # SNN configuration (SELU, no normalisation): identity statistics and affine
        dev = y.device
        mr = torch.tensor([[0.0, 1.0]] * y.shape[0], device=dev)
        g, bt = torch.ones(y.shape[4], device=dev), torch.zeros(y.shape[4], device=dev)
    else:
        g, bt = norm.weight, norm.bias
    if residual is None:
        return GNActFn.apply(y, mr, g, bt, act, None, None, None, None, nstat, 0, eps)
    y2, mr2, norm2, nstat2 = residual
    if norm2 is None:
        dev = y.device
        mr2 = torch.tensor([[0.0, 1.0]] * y.shape[0], device=dev)
        g2, b2 = torch.ones(y.shape[4], device=dev), torch.zeros(y.shape[4], device=dev)
    else:
        g2, b2 = norm2.weight, norm2.bias
    return GNActFn.apply(y, mr, g, bt, act, y2, mr2, g2, b2, nstat, nstat2, eps)


class CropHighFn(torch.autograd.Function):
    """x[:, lo0:lo0 + d, lo1:lo1 + h, lo2:lo2 + w, :] of a channels-last tensor as ONE node: a Python slice over three axes is three
    autograd nodes whose backward each fills and copies a full-size tensor (6 kernels per decoder level and step in the V-Net trace);
    here the backward is one zero fill and one copy."""

    @staticmethod
    def forward(ctx, x, lo, size):
        ctx.lo, ctx.size, ctx.shape = tuple(lo), tuple(size), tuple(x.shape)
        (a, b, c), (d, h, w) = ctx.lo, ctx.size
        return x[:, a:a + d, b:b + h, c:c + w, :].contiguous()

    @staticmethod
    def backward(ctx, g):
        (a, b, c), (d, h, w) = ctx.lo, ctx.size
        gx = torch.zeros(ctx.shape, device=g.device, dtype=g.dtype)
        gx[:, a:a + d, b:b + h, c:c + w, :] = g
        return gx, None, None


class LegWeightsFn(torch.autograd.Function):
    """The (K, C_total) weight of the deep-supervision 1x1x1 convolution over the concatenated legs (nets/architectures.py:474-476),
    handed out as one contiguous (8, C_leg) matrix per leg (K rows padded to the 8 the bf16 GEMM needs).  One node: per-leg column
    slices + F.pad were ~3 kernels per leg forward and, through autograd's slice / pad backward, a zero-filled (K, C_total) tensor
    and an accumulation per leg backward (14 launches); the backward here is one concatenation."""

    @staticmethod
    def forward(ctx, w, sizes):
        K, Ct = w.shape
        ctx.K = K
        wp = torch.zeros((K + (-K) % 8, Ct), device=w.device, dtype=w.dtype)
        wp[:K] = w
        out, c0 = [], 0
        for c in sizes:
            out.append(wp[:, c0:c0 + c].contiguous())
            c0 += c
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        return torch.cat(grads, dim=1)[:ctx.K], None


def pointwise_to_f32(x, weight, bias, out_channels, padded=False):
    """padded: `weight` already has its rows padded to a multiple of 8 (LegWeightsFn)"""
    if padded:
        b = torch.nn.functional.pad(bias, (0, weight.shape[0] - bias.shape[0])) if bias is not None else None
        y, _ = ConvFn.apply(x, None, weight.reshape(weight.shape[0], weight.shape[1], 1, 1, 1), b, 1, 1, False, False, 1e-5)
        return UnpackFn.apply(y, out_channels)
    return _pointwise_to_f32(x, weight, bias, out_channels)


def _pointwise_to_f32(x, weight, bias, out_channels):
    """1x1x1 conv of a channels-last bf16 tensor to a few (< 8) output channels, returned as fp32 NCDHW: where the bf16
    body hands over to the fp32 head (deep-supervision legs, conv_out).  The output channels are padded to 8 for the GEMM."""
    w2 = weight.reshape(weight.shape[0], -1)
    pad_o = (-w2.shape[0]) % 8
    W = torch.nn.functional.pad(w2, (0, 0, 0, pad_o)).reshape(w2.shape[0] + pad_o, w2.shape[1], 1, 1, 1)
    b = torch.nn.functional.pad(bias, (0, pad_o)) if bias is not None else None
    y, _ = ConvFn.apply(x, None, W, b, 1, 1, False, False, 1e-5)
    return UnpackFn.apply(y, out_channels)
