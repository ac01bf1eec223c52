"""FNO / FNOSeg / HNOSeg (NeuralOperatorSeg), HartleyMHASeg and V-Net-DS on the HIP kernels
(reference nets/architectures.py:26-653).  Same constructors, module tree and state-dict keys."""
from functools import partial
from typing import Union

import numpy as np
import torch
from torch import nn

from .. import ops
from .fourier_operator import FourierOperator
from .hartley_operator import HartleyOperator
from .nets_utils import ConvNormAct, ConvTransposeNormAct, init_weights_for_snn, spatial_padcrop, _is_selu


class _TransBlock(nn.Module):
    """x1 = op(x); x2 = conv_branch(x); x = act(norm(x1 + x2)); block skip (reference :511-548)."""

    def __init__(self):
        super().__init__()
        self.op = self.conv_branch = self.normalization = self.activation = None
        self.use_block_skip = None
        self.conv_concat = None

    def forward(self, x):
        act = ops.act_id(self.activation)
        fuse_act = act if self.normalization is None else ops.ACT_NONE
        x2 = None
        if self.conv_branch is not None:
            x2 = ops.PwConvFn.apply(x, None, self.conv_branch.weight, self.conv_branch.bias, ops.ACT_NONE)
        assert self.op is not None or x2 is not None
        if self.op is not None:
            # the operator adds the conv branch and applies the activation on store
            y = self.op.forward_fused(x, addend=x2, act=fuse_act)
        else:
            from .elementwise import activation_forward
            y = activation_forward(x2, fuse_act)
        if self.normalization is not None:
            from .conv3d import group_norm_act
            y = group_norm_act(y, self.normalization, act)
        if self.use_block_skip:
            if self.conv_concat is not None:
                return self.conv_concat(y, x)
            return y + x
        return y


class _Pending(nn.Module):
    """Placeholder until the corresponding HIP path lands (later rows of SURVEY.md section 8)."""
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(f'{type(self).__name__} is not provided by the HIP path yet')


class NeuralOperatorSeg(_Pending):
    pass


class HartleyMHASeg(_Pending):
    pass


class VNetDS(_Pending):
    pass
