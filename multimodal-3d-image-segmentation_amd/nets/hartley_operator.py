"""HartleyOperator on the HIP kernels (reference nets/hartley_operator.py:17-333)."""
import math

import numpy as np
import torch
from torch.nn import Module, Parameter, init

from .. import ops


def _add_at_origin(y, per_channel):
    """y[b, o, 0, 0, 0] += per_channel[o] (differentiable in both arguments)."""
    onehot = torch.zeros(y.shape[2:], device=y.device, dtype=y.dtype)
    onehot[(0,) * onehot.ndim] = 1.0
    return y + per_channel.reshape(1, -1, *([1] * (y.ndim - 2))) * onehot


class HartleyOperator(Module):
    """Frequency-domain channel mixing through the Hartley transform.

    Same constructor, parameters (`weight` (Co,Ci) or (Co,Ci,2m0,2m1,2m2), optional `bias`
    (1,Co,1,1,1)) and semantics as the reference: with ``use_transform`` the input is
    transformed, mixed on the kept mode block, SELU is applied IN THE FREQUENCY DOMAIN and the
    unscaled transform brings it back (reference :168-271); without it the input already is a
    cropped spectrum (HNOSeg-XS, reference :287-299).
    """

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
            shape = shape + tuple(2 * m for m in self.num_modes)
        self.weight = Parameter(torch.empty(shape, device=device, dtype=dtype))
        if use_bias:
            self.bias = Parameter(torch.empty((1, out_channels) + (1,) * (ndim - 2), device=device, dtype=dtype))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            init.zeros_(self.bias)

    # -- forward -----------------------------------------------------------------------
    def _is_lifted_2d(self, inputs):
        """a 2-D operator (two mode counts / 4-D per-mode weights) fed the (B, C, 1, H, W) view of a 2-D model"""
        return inputs.ndim == 5 and (self.weight.ndim == 4 or (self.num_modes is not None and len(self.num_modes) == 2))

    def forward(self, inputs):
        if inputs.ndim == 4:
            return self._lifted()(inputs.unsqueeze(2)).squeeze(2)
        if self._is_lifted_2d(inputs):
            return self._lifted()(inputs)
        if self.use_transform:
            return self._call3d(inputs)
        x = self._mix(inputs)
        if self.use_bias:
            x = x + self.bias
        return x

    def _lifted(self):
        """2-D operator (reference _call2d :108-166, _call2d_notransform :273-285) as the 3-D one on
        (B, C, 1, H, W): a size-1 axis has the single frequency 0, its reversal is the identity and its
        transform is a copy, so the 3-D kernels compute exactly the 2-D result."""
        view = object.__new__(HartleyOperator)
        Module.__init__(view)
        view.in_channels, view.out_channels = self.in_channels, self.out_channels
        view.use_bias, view.weights_type, view.use_transform = self.use_bias, self.weights_type, self.use_transform
        view.num_modes = None if self.num_modes is None else (0,) + tuple(self.num_modes)
        view.__dict__['weight'] = self.weight if self.weights_type == 'shared' else self.weight.unsqueeze(2)
        view.__dict__['bias'] = None if self.bias is None else self.bias.unsqueeze(2)
        return view

    def _mix(self, z):
        if self.weights_type == 'shared':
            return ops.PwConvFn.apply(z, None, self.weight, None, ops.ACT_NONE)   # 'oi,bidhw->bodhw'
        from .spectral_individual import hartley_mix_individual
        return hartley_mix_individual(z, self.weight)

    def forward_fused(self, inputs, addend=None, act=ops.ACT_NONE):
        """act(self(inputs) + addend) with the add and the activation fused into the inverse transform's
        store (the block pattern of nets/architectures.py:521-539)."""
        if inputs.ndim == 4:
            return self._lifted().forward_fused(inputs.unsqueeze(2), None if addend is None else addend.unsqueeze(2),
                                                act).squeeze(2)
        if self._is_lifted_2d(inputs):
            return self._lifted().forward_fused(inputs, addend, act)
        if not self.use_transform or addend is None and act == ops.ACT_NONE:
            y = self(inputs)
            if addend is not None:
                y = ops.AddFn.apply(y, addend)
            return ops.ActFn.apply(y, act) if act != ops.ACT_NONE else y
        return self._call3d(inputs, addend, act)

    def _call3d(self, inputs, addend=None, act=ops.ACT_NONE):
        spatial = tuple(inputs.shape[2:])
        modes = self.num_modes
        if self.weights_type == 'shared':
            modes = ops.clamp_modes(modes, spatial)
        else:
            assert all(s >= 2 * m for s, m in zip(spatial, modes))
        n3 = float(np.prod(spatial))
        bias = self.bias.reshape(-1) if self.use_bias else None
        z = ops.DhtCropFn.apply(inputs, modes, 1.0 / n3) if self.weights_type == 'shared' else None
        if self.weights_type == 'shared':
            # selu(0) = 0, so SELU on the padded spectrum == SELU on the kept block (without bias)
            z = ops.PwConvFn.apply(z, None, self.weight, bias, ops.ACT_SELU)
        else:
            from .spectral_individual import hartley_mix_individual
            if bias is None:
                z = hartley_mix_individual(None, self.weight, act=ops.ACT_SELU, x_full=inputs, modes=modes)
            else:
                z = hartley_mix_individual(None, self.weight, act=ops.ACT_NONE, x_full=inputs, modes=modes)
                z = ops.BiasActFn.apply(z, self.bias, ops.ACT_SELU)
        delta = None
        if bias is not None:
            # The reference adds the bias to the ZERO-PADDED spectrum (hartley_operator.py:262-263), so after SELU the
            # whole padded region holds the constant selu(b).  The (unscaled) inverse transform of a constant spectrum is
            # a delta at the origin, N^3 * selu(b): subtract the constant on the kept block, add the delta afterwards.
            cb = torch.nn.functional.selu(self.bias)                       # (1, Co, 1, 1, 1): Co values of a parameter
            z = ops.BiasActFn.apply(z, -cb, ops.ACT_NONE)                  # z - selu(b) on the kept block
            delta = cb.reshape(-1) * n3
        if delta is not None:   # rare switch: unfused tail so that the delta lands before the residual add / activation
            y = ops.PadIdhtFn.apply(z, spatial, 1.0, ops.ACT_NONE)
            y = _add_at_origin(y, delta)
            if addend is not None:
                y = ops.AddFn.apply(y, addend)
            return ops.ActFn.apply(y, act) if act != ops.ACT_NONE else y
        if addend is None:
            return ops.PadIdhtFn.apply(z, spatial, 1.0, act)
        return ops.PadIdhtAddFn.apply(z, addend, spatial, 1.0, act)


def get_reverse(x, dims):
    """x[N - k] along `dims` by "flip, then roll by one" (reference nets/hartley_operator.py:320-333).  Pure index
    permutation (no arithmetic)."""
    assert isinstance(dims, (list, tuple))
    return torch.roll(torch.flip(x, dims), [1] * len(dims), dims)


def hartley_conv(equation, weight, weight_reverse, x, x_reverse):
    """Hartley convolution theorem in the frequency domain (reference nets/hartley_operator.py:302-317):

        1/2 [ einsum(equation, weight, x + x_reverse) + einsum(equation, weight_reverse, x - x_reverse) ]

    for the equations the reference uses -- 'oi,bi...->bo...' (one matrix for all modes) and 'oi...,bi...->bo...'
    (one matrix per mode), 2-D or 3-D.  The two combinations are formed by hno_axpby (with the 1/2 folded in); the shared
    form is then ONE concat-fused pointwise conv [W | W_rev] . [s ; d], the per-mode form one complex per-mode mix
    (W + i W_rev)(s + i d') whose real part is W s - W_rev d' with d' = (x_reverse - x)/2."""
    lhs, rhs = equation.replace(' ', '').split('->')[0].split(',')
    nsp = len(rhs) - 2
    if nsp not in (2, 3) or len(lhs) not in (2, 2 + nsp) or x.ndim != nsp + 2:
        raise ValueError(f'hartley_conv: unsupported equation {equation!r}')
    if nsp == 2:        # 2-D: the same kernels on a (B, C, 1, H, W) view
        eq3 = 'oi,bidhw->bodhw' if len(lhs) == 2 else 'oidhw,bidhw->bodhw'
        lift = (lambda w: w) if len(lhs) == 2 else (lambda w: w.unsqueeze(2))
        return hartley_conv(eq3, lift(weight), lift(weight_reverse), x.unsqueeze(2), x_reverse.unsqueeze(2)).squeeze(2)
    s = ops.AxpbyFn.apply(x, x_reverse, 0.5, 0.5)
    if len(lhs) == 2:
        d = ops.AxpbyFn.apply(x, x_reverse, 0.5, -0.5)
        return ops.PwConvFn.apply(s, d, torch.cat([weight, weight_reverse], dim=1), None, ops.ACT_NONE)
    d = ops.AxpbyFn.apply(x_reverse, x, 0.5, -0.5)
    y = ops.PerModeFourierFn.apply(torch.cat([s, d], dim=1), weight, weight_reverse)      # [re | im] channel halves
    return y[:, :weight.shape[0]]
