"""Layer helpers with the reference's names and state-dict layout (nets/nets_utils.py).

nn.Conv3d / nn.GroupNorm instances are used purely as PARAMETER CONTAINERS (same default
initialisation and RNG consumption as the reference, same ``op.weight`` / ``op.bias`` keys);
their ATen forward is never called -- the compute goes through the HIP ops.
"""
from typing import List

import numpy as np
import torch
from torch import nn

from .. import ops


def get_spatial_padcrop(x: torch.Tensor, target_shape: List[int]):
    """Padding / cropping tuples in F.pad order (last axis first); odd differences put the
    extra element on the high side (reference nets/nets_utils.py:60-99)."""
    shape = tuple(x.shape[2:])
    nd = len(shape)
    pad, crop = [0, 0] * nd, [0, 0] * nd
    for i, (t, s) in enumerate(zip(reversed(tuple(target_shape)), reversed(shape))):
        d = t - s
        lo = abs(d) // 2
        tgt = pad if d >= 0 else crop
        tgt[2 * i], tgt[2 * i + 1] = lo, abs(d) - lo
    return pad, crop


def spatial_padcrop(x: torch.Tensor, target_shape: List[int]):
    """Centre pad and/or crop to target_shape; no-op when shapes match
    (reference nets/nets_utils.py:22-57).  Pure indexing, no arithmetic."""
    nd = x.ndim - 2
    assert x.ndim in (3, 4, 5) and nd == len(target_shape)
    if tuple(x.shape[2:]) == tuple(target_shape):
        return x
    pad, crop = get_spatial_padcrop(x, target_shape)
    if any(pad):
        x = nn.functional.pad(x, pad)
    if any(crop):
        idx = [slice(None), slice(None)]
        for ax in range(nd):
            lo, hi = crop[2 * (nd - 1 - ax)], crop[2 * (nd - 1 - ax) + 1]
            idx.append(slice(lo, x.shape[2 + ax] - hi))
        x = x[tuple(idx)]
    return x


def init_weights_for_snn(module):
    """SNN initialisation (reference nets/nets_utils.py:102-117): kaiming-normal with linear
    gain on conv and spectral weights, bias ~ U(-1e-3, 1e-3).  HartleyMultiHeadAttention is
    deliberately NOT in the target list, as in the reference."""
    from .fourier_operator import FourierOperator
    from .hartley_operator import HartleyOperator
    if isinstance(module, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d, HartleyOperator)):
        nn.init.kaiming_normal_(module.weight, nonlinearity='linear')
        if module.bias is not None:
            nn.init.uniform_(module.bias, -0.001, 0.001)
    elif isinstance(module, FourierOperator):
        nn.init.kaiming_normal_(module.weight_real, nonlinearity='linear')
        nn.init.kaiming_normal_(module.weight_imag, nonlinearity='linear')
        if module.bias is not None:
            nn.init.uniform_(module.bias, -0.001, 0.001)


def _is_selu(activation):
    return activation == 'selu' or activation is nn.functional.selu


def _lib_error(msg):
    from .. import _lib
    return _lib.HnoError(msg)


def conv_forward(op, x, act_id, xb=None):
    """Run a conv parameter container through the HIP kernels (optionally with a fused concat
    of a second input for 1x1x1 convs)."""
    nsp = op.weight.ndim - 2                       # 2 for nn.Conv2d parameters, 3 for nn.Conv3d
    k = op.kernel_size if not np.isscalar(op.kernel_size) else (op.kernel_size,) * nsp
    s = op.stride if not np.isscalar(op.stride) else (op.stride,) * nsp
    if x.ndim != 5:
        raise _lib_error('conv_forward expects (B, C, D, H, W); 2-D models lift their input to D = 1 first')
    if all(v == 1 for v in k) and all(v == 1 for v in s):
        return ops.PwConvFn.apply(x, xb, op.weight, op.bias, act_id)
    assert xb is None
    if all(v == 2 for v in k) and all(v == 2 for v in s):
        w = op.weight
        if nsp == 2:
            # 2-D model on a (B, C, 1, H, W) view: Conv2d(k 2, s 2, p 1) == the 3-D kernel with the 2-D taps at kd = 1 (the
            # kd = 0 taps read the depth padding); autograd slices the weight gradient back
            assert x.shape[2] == 1
            w = torch.stack([torch.zeros_like(w), w], dim=2)
        if w.shape[0] > 32 or w.shape[1] > 8 or (x.requires_grad and torch.is_grad_enabled()):
            # beyond the fast kernel's limits (<= 32 output, <= 8 input channels, no input gradient): the direct kernels (round 6) --
            # the reference takes any `filters` / any number of modalities (nets/hnosegxs.py:46-62)
            y = ops.ConvKFn.apply(x, w, op.bias, 2, False, 1)
            return ops.ActFn.apply(y, act_id) if act_id != ops.ACT_NONE else y
        return ops.ConvK2S2Fn.apply(x, w, op.bias, act_id)
    from .conv3d import conv3d_forward  # general kernels (V-Net path)
    return conv3d_forward(op, x, act_id)


class _OpNormAct(nn.Module):
    def __init__(self):
        super().__init__()
        self.op = None
        self.normalization = None
        self.activation = None

    def forward(self, x, xb=None):
        """y = act(norm(op([x ; xb]))).  `xb` is an optional second tensor concatenated along
        channels inside the kernel (replaces torch.cat at the call sites)."""
        if self.normalization is None:
            return conv_forward(self.op, x, ops.act_id(self.activation), xb)
        from .conv3d import group_norm_act
        y = conv_forward(self.op, x, ops.ACT_NONE, xb)
        return group_norm_act(y, self.normalization, ops.act_id(self.activation))


class ConvNormAct(_OpNormAct):
    """conv -> [GroupNorm(1, C)] -> activation (reference nets/nets_utils.py:136-174)."""

    def __init__(self, in_channels, out_channels, *, kernel_size=1, stride=1, use_bias=True, activation='selu',
                 use_snn=True, ndim=5, device=None):
        super().__init__()
        assert ndim in (4, 5)
        if np.all(np.array(stride) == 1):
            padding = 'same'
        else:
            padding = kernel_size // 2 if np.isscalar(kernel_size) else tuple(np.array(kernel_size) // 2)
        conv = nn.Conv2d if ndim == 4 else nn.Conv3d
        self.op = conv(in_channels, out_channels, kernel_size, stride, padding, bias=use_bias, device=device)
        if use_snn:
            if not _is_selu(activation):
                raise RuntimeError('Self-normalizing neural network (SNN) must be used with SELU.')
        else:
            self.normalization = nn.GroupNorm(1, out_channels, device=device)
        self.activation = getattr(nn.functional, activation) if isinstance(activation, str) else activation


class ConvTransposeNormAct(_OpNormAct):
    """ConvTranspose(stride 2, padding k//2, output_padding 1) -> [GroupNorm] -> activation
    (reference nets/nets_utils.py:177-211)."""

    def __init__(self, in_channels, out_channels, *, kernel_size=2, use_bias=True, activation='selu', ndim=5,
                 device=None):
        super().__init__()
        assert ndim in (4, 5)
        padding = kernel_size // 2 if np.isscalar(kernel_size) else tuple(np.array(kernel_size) // 2)
        conv = nn.ConvTranspose2d if ndim == 4 else nn.ConvTranspose3d
        self.op = conv(in_channels, out_channels, kernel_size, 2, padding, 1, bias=use_bias, device=device)
        if not _is_selu(activation):
            self.normalization = nn.GroupNorm(1, out_channels, device=device)
        self.activation = getattr(nn.functional, activation) if isinstance(activation, str) else activation

    def forward(self, x, xb=None):
        from .conv3d import conv_transpose3d_forward, group_norm_act
        assert xb is None
        if self.normalization is None:
            return conv_transpose3d_forward(self.op, x, ops.act_id(self.activation))
        y = conv_transpose3d_forward(self.op, x, ops.ACT_NONE)
        return group_norm_act(y, self.normalization, ops.act_id(self.activation))
