"""FourierOperator on the HIP kernels (reference nets/fourier_operator.py:15-223)."""
import math

import numpy as np
import torch
from torch.nn import Module, Parameter, init

from .. import ops


class FourierOperator(Module):
    """rfftn(norm='forward') -> complex channel mix on the kept corners -> zero pad ->
    irfftn(norm='forward') (unscaled).  Parameters `weight_real` / `weight_imag` are
    (Co,Ci) ('shared') or (Co,Ci,2m0,2m1,m2) ('individual'), as in the reference (:67-76)."""

    def __init__(self, in_channels, out_channels, num_modes=None, use_bias=False, weights_type='shared',
                 use_transform=True, ndim=5, device=None, dtype=None):
        super().__init__()
        valid = {'individual', 'shared'}
        if weights_type not in valid:
            raise ValueError(f'weights_type must be one of {valid}')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_bias, self.weights_type, self.use_transform = use_bias, weights_type, use_transform
        self.num_modes = num_modes
        if num_modes is not None:
            if np.isscalar(num_modes):
                self.num_modes = (num_modes,) * (ndim - 2)
            else:
                assert len(num_modes) == ndim - 2
                self.num_modes = tuple(num_modes)
        shape = (out_channels, in_channels)
        if weights_type != 'shared':
            assert self.num_modes is not None
            shape = shape + tuple(2 * m for m in self.num_modes[:-1]) + (self.num_modes[-1],)
        self.weight_real = Parameter(torch.empty(shape, device=device, dtype=dtype))
        self.weight_imag = Parameter(torch.empty(shape, device=device, dtype=dtype))
        if use_bias:
            self.bias = Parameter(torch.empty((1, out_channels) + (1,) * (ndim - 2), device=device, dtype=dtype))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight_real, a=math.sqrt(5))
        init.kaiming_uniform_(self.weight_imag, a=math.sqrt(5))
        if self.bias is not None:
            init.zeros_(self.bias)

    def forward(self, inputs):
        from .spectral_fourier import fourier_operator_forward
        return fourier_operator_forward(self, inputs)

    def forward_fused(self, inputs, addend=None, act=ops.ACT_NONE):
        """act(self(inputs) + addend), add and activation fused into the inverse transform's store."""
        from .spectral_fourier import fourier_operator_forward
        return fourier_operator_forward(self, inputs, addend, act)
