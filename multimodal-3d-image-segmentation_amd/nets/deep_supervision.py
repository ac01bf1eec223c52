"""conv_ds over the channel concat of all block outputs (reference nets/architectures.py:341-343)
without materialising the concat: act(sum_t W[:, slice_t] x_t + b)."""
from .. import ops


def conv_over_concat(conv_norm_act, tensors):
    op = conv_norm_act.op
    w = op.weight.reshape(op.weight.shape[0], -1)
    acc, c0 = None, 0
    for i, t in enumerate(tensors):
        c = t.shape[1]
        part = ops.PwConvFn.apply(t, None, w[:, c0:c0 + c].contiguous(), op.bias if i == 0 else None, ops.ACT_NONE)
        acc = part if acc is None else ops.AddFn.apply(acc, part)
        c0 += c
    act = ops.act_id(conv_norm_act.activation)
    if conv_norm_act.normalization is not None:      # non-SELU activations: conv -> GroupNorm(1, C) -> activation (nets_utils.py:127-133)
        from .conv3d import group_norm_act
        return group_norm_act(acc, conv_norm_act.normalization, act)
    return ops.ActFn.apply(acc, act) if act != ops.ACT_NONE else acc
