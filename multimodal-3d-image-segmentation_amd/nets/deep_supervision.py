"""conv_ds over the channel concat of all block outputs (reference nets/architectures.py:341-343)
without materialising the concat: act(sum_t W[:, slice_t] x_t + b)."""
import torch

from .. import ops


class _LegWeights(torch.autograd.Function):
    """The (K, T * C) weight of conv_ds handed out as T contiguous (K, C) matrices (separate outputs of ONE node: indexing a stacked
    tensor would put a SelectBackward -- zero fill, copy, accumulate -- behind every leg) -- one permuting copy forward, one stack backward.
    Per-leg column slices + .contiguous() were a clone and a copy per leg forward and, through autograd's SliceBackward, a zero-filled
    (K, T * C) tensor, a copy and an accumulation per leg backward: ~100 tiny launches per HartleyMHASeg step (17 legs; round 6 census)."""

    @staticmethod
    def forward(ctx, w, T):
        K = w.shape[0]
        ctx.shape = (K, w.shape[1])
        ctx.T = T
        return tuple(w.reshape(K, T, -1).permute(1, 0, 2).contiguous().unbind(0))            # T x (K, C), one buffer

    @staticmethod
    def backward(ctx, *grads):
        like = next(g for g in grads if g is not None)
        g = torch.stack([torch.zeros_like(like) if gi is None else gi for gi in grads], dim=0)
        return g.permute(1, 0, 2).reshape(ctx.shape), None


def conv_over_concat(conv_norm_act, tensors):
    op = conv_norm_act.op
    w = op.weight.reshape(op.weight.shape[0], -1)
    chans = [t.shape[1] for t in tensors]
    same = len(set(chans)) == 1 and not w.is_meta
    import os
    if (same and len(tensors) > 1 and os.environ.get('HNO_DS_MULTI', '1') != '0' and ops.MultiPwConvFn.supported(tensors, w.shape[0])):
        # round 6: all legs in ONE launch each way (csrc/hno_pwmulti.hip) instead of a pointwise convolution + an add per leg; the
        # (K, T C) weight regrouped by leg -- (T, K, C) -- by one permuting copy (and one back for its gradient)
        K, T = w.shape[0], len(tensors)
        w_tkc = w.reshape(K, T, chans[0]).permute(1, 0, 2).contiguous()
        acc = ops.MultiPwConvFn.apply(w_tkc, op.bias, *tensors)
        act = ops.act_id(conv_norm_act.activation)
        if conv_norm_act.normalization is not None:
            from .conv3d import group_norm_act
            return group_norm_act(acc, conv_norm_act.normalization, act)
        return ops.ActFn.apply(acc, act) if act != ops.ACT_NONE else acc
    legs = _LegWeights.apply(w, len(tensors)) if same else None
    acc, c0 = None, 0
    for i, t in enumerate(tensors):
        c = chans[i]
        wt = legs[i] if same else w[:, c0:c0 + c].contiguous()
        part = ops.PwConvFn.apply(t, None, wt, op.bias if i == 0 else None, ops.ACT_NONE)
        acc = part if acc is None else ops.AddFn.apply(acc, part)
        c0 += c
    act = ops.act_id(conv_norm_act.activation)
    if conv_norm_act.normalization is not None:      # non-SELU activations: conv -> GroupNorm(1, C) -> activation (nets_utils.py:127-133)
        from .conv3d import group_norm_act
        return group_norm_act(acc, conv_norm_act.normalization, act)
    return ops.ActFn.apply(acc, act) if act != ops.ACT_NONE else acc
