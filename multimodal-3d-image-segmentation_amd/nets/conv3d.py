"""Dispatch of the V-Net-DS layer types to the HIP kernels (3x3x3 convolutions, transposed convolution,
GroupNorm(1, C) + activation)."""
import numpy as np
import torch

from .. import ops


def _k(op):
    return tuple(op.kernel_size) if not np.isscalar(op.kernel_size) else (op.kernel_size,) * 3


def _embed_2d(w):
    """(C0, C1, k, k) weights of a 2-D layer -> (C0, C1, k, k, k) with the 2-D taps in the middle depth plane: on a
    (B, C, 1, H, W) view with padding k // 2 the outer depth taps only ever meet the padding (autograd slices the gradient back)."""
    k = w.shape[-1]
    z = torch.zeros_like(w)
    return torch.stack([z] * (k // 2) + [w] + [z] * (k // 2), dim=2)


def conv3d_forward(op, x, act_id):
    """nn.Conv3d parameter container -> implicit-GEMM kernel (kernel 3, stride 1 'same' or stride 2 padding 1).  nn.Conv2d
    containers (2-D models on a D = 1 view) go through the same kernel with embedded weights."""
    nsp = op.weight.ndim - 2
    k = tuple(op.kernel_size) if not np.isscalar(op.kernel_size) else (op.kernel_size,) * nsp
    s = tuple(op.stride) if not np.isscalar(op.stride) else (op.stride,) * nsp
    if len(set(k)) != 1 or k[0] % 2 == 0 or s not in ((1,) * nsp, (2,) * nsp):
        raise NotImplementedError(f'Conv kernel {k} stride {s} is not provided by the HIP path (cubic kernels of odd size at stride 1 | 2; 2x2x2/s2)')
    w = op.weight
    if nsp == 2:
        assert x.shape[2] == 1
        w = _embed_2d(w)
    if k[0] == 3:
        y = ops.Conv3dK3Fn.apply(x, w, op.bias, s[0])
    else:       # the reference's `kernel_size` argument (nets/architectures.py:55-70): any odd size through the direct kernels (round 6)
        y = ops.ConvKFn.apply(x, w, op.bias, s[0], False)
    return ops.ActFn.apply(y, act_id) if act_id != ops.ACT_NONE else y


def conv_transpose3d_forward(op, x, act_id):
    nsp = op.weight.ndim - 2
    k = tuple(op.kernel_size) if not np.isscalar(op.kernel_size) else (op.kernel_size,) * nsp
    if len(set(k)) != 1 or k[0] % 2 == 0:
        raise NotImplementedError(f'ConvTranspose kernel {k} is not provided by the HIP path (cubic kernels of odd size, stride 2)')
    w = op.weight
    fn = (lambda xx, ww, bb: ops.ConvT3dK3Fn.apply(xx, ww, bb)) if k[0] == 3 else (lambda xx, ww, bb: ops.ConvKFn.apply(xx, ww, bb, 2, True))
    if nsp == 2:
        # ConvTranspose2d on the D = 1 view: the 3-D kernel doubles the depth as well; with the taps in the middle depth plane
        # output plane 0 is the 2-D result (plane 1 only holds the bias) -- keep plane 0
        assert x.shape[2] == 1
        y = fn(x, _embed_2d(w), op.bias)[:, :, :1].contiguous()
    else:
        y = fn(x, w, op.bias)
    return ops.ActFn.apply(y, act_id) if act_id != ops.ACT_NONE else y


def group_norm_act(y, norm, act_id):
    assert norm.num_groups == 1
    return ops.GroupNormActFn.apply(y, norm.weight, norm.bias, norm.eps, act_id)
