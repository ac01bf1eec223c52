"""Dispatch of the V-Net-DS layer types to the HIP kernels (3x3x3 convolutions, transposed convolution,
GroupNorm(1, C) + activation)."""
import numpy as np

from .. import ops


def _k(op):
    return tuple(op.kernel_size) if not np.isscalar(op.kernel_size) else (op.kernel_size,) * 3


def conv3d_forward(op, x, act_id):
    """nn.Conv3d parameter container -> implicit-GEMM kernel (kernel 3, stride 1 'same' or stride 2 padding 1)."""
    k, s = _k(op), tuple(op.stride) if not np.isscalar(op.stride) else (op.stride,) * 3
    if k != (3, 3, 3) or s not in ((1, 1, 1), (2, 2, 2)):
        raise NotImplementedError(f'Conv3d kernel {k} stride {s} is not provided by the HIP path (1x1x1, 2x2x2/s2, 3x3x3/s1|s2)')
    y = ops.Conv3dK3Fn.apply(x, op.weight, op.bias, s[0])
    return ops.ActFn.apply(y, act_id) if act_id != ops.ACT_NONE else y


def conv_transpose3d_forward(op, x, act_id):
    k = _k(op)
    if k != (3, 3, 3):
        raise NotImplementedError(f'ConvTranspose3d kernel {k} is not provided by the HIP path (3x3x3, stride 2)')
    y = ops.ConvT3dK3Fn.apply(x, op.weight, op.bias)
    return ops.ActFn.apply(y, act_id) if act_id != ops.ACT_NONE else y


def group_norm_act(y, norm, act_id):
    assert norm.num_groups == 1
    return ops.GroupNormActFn.apply(y, norm.weight, norm.bias, norm.eps, act_id)
