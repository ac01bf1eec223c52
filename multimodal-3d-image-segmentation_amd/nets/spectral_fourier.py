"""FourierOperator forward on the HIP kernels (reference nets/fourier_operator.py:90-223).

The kept half spectrum is carried as REAL data (B, 2C, 2m0, 2m1, m2) with channels [re | im], so
the complex channel mix  Y = (Wr + i Wi) X  is ONE real pointwise convolution over the mode axis:
    [Yr ; Yi] = [[Wr, -Wi], [Wi, Wr]] [Xr ; Xi]
(the same MFMA kernels as every other 1x1x1 conv; ops.ComplexMixFn composes the matrix and splits its
gradient back into weight_real / weight_imag with two tiny kernels).
"""
import numpy as np
import torch

from .. import ops


def complex_mix_shared(spec, weight_real, weight_imag):
    return ops.ComplexMixFn.apply(spec, weight_real, weight_imag)


def fourier_operator_forward(op, inputs, addend=None, act=ops.ACT_NONE):
    lifted = inputs.ndim == 5 and (op.weight_real.ndim == 4 or (op.num_modes is not None and len(op.num_modes) == 2))
    if inputs.ndim == 4 or lifted:
        # 2-D (reference _call2d :117-160): the same kernels on (B, C, 1, H, W) -- the transform of a size-1 axis is a copy
        # (`lifted`: a 2-D operator inside a 2-D model that already runs on that view)
        from types import SimpleNamespace
        ind = op.weights_type != 'shared'
        view = SimpleNamespace(use_transform=op.use_transform, weights_type=op.weights_type, use_bias=op.use_bias,
                               num_modes=None if op.num_modes is None else (0,) + tuple(op.num_modes),
                               weight_real=op.weight_real.unsqueeze(2) if ind else op.weight_real,
                               weight_imag=op.weight_imag.unsqueeze(2) if ind else op.weight_imag,
                               bias=None if op.bias is None else op.bias.unsqueeze(2))
        if lifted:
            return fourier_operator_forward(view, inputs, addend, act)
        y = fourier_operator_forward(view, inputs.unsqueeze(2), None if addend is None else addend.unsqueeze(2), act)
        return y.squeeze(2)
    if not op.use_transform:
        # complex spectrum in, complex spectrum out (fourier_operator.py:97-105, 212-223): the mix runs on the real
        # [re | im] channel layout of the kernels; the bias is real and added to the result
        assert addend is None and act == ops.ACT_NONE
        if not inputs.is_complex():
            raise ValueError('FourierOperator(use_transform=False) expects a complex input')
        spec = torch.cat([inputs.real, inputs.imag], dim=1).float().contiguous()
        if op.weights_type == 'shared':
            y = complex_mix_shared(spec, op.weight_real, op.weight_imag)
        else:
            y = ops.PerModeFourierFn.apply(spec, op.weight_real, op.weight_imag)
        co = y.shape[1] // 2
        yr, yi = y[:, :co], y[:, co:]
        if op.use_bias:
            yr = yr + op.bias
        return torch.complex(yr.contiguous(), yi.contiguous())
    spatial = tuple(inputs.shape[2:])
    modes = ops.clamp_modes(op.num_modes, spatial) if op.weights_type == 'shared' else tuple(op.num_modes)
    spec = ops.RfftCropFn.apply(inputs, modes)
    if op.weights_type == 'shared':
        spec = complex_mix_shared(spec, op.weight_real, op.weight_imag)
    else:
        assert all(s >= 2 * m for s, m in zip(spatial, op.num_modes))
        spec = ops.PerModeFourierFn.apply(spec, op.weight_real, op.weight_imag)
    if not op.use_bias:
        return ops.IrfftPadFn.apply(spec, addend, spatial, act)
    # The reference adds the (real) bias to the padded half spectrum, i.e. at every (k0, k1) and k2 < m2
    # (fourier_operator.py:206-207), before the unscaled irfftn.  A constant over (k0, k1) is a delta at
    # (n0, n1) = (0, 0) times N0 N1; along the last axis the c2r transform of [b] * m2 is b * (1 + 2 sum_k cos).
    N0, N1, N2 = spatial
    m2 = modes[2]
    n2 = torch.arange(N2, device=inputs.device, dtype=torch.float64)
    k = torch.arange(1, m2, device=inputs.device, dtype=torch.float64)
    line = (1.0 + 2.0 * torch.cos(2.0 * np.pi * k[:, None] * n2[None, :] / N2).sum(0)).float() * float(N0 * N1)   # (N2,)
    y = ops.IrfftPadFn.apply(spec, None, spatial, ops.ACT_NONE)
    term = op.bias.reshape(1, -1, 1) * line.reshape(1, 1, -1)                                   # (1, Co, N2)
    mask = torch.zeros((N0, N1, 1), device=inputs.device, dtype=y.dtype)
    mask[0, 0, 0] = 1.0
    y = y + term[:, :, None, None, :] * mask
    if addend is not None:
        y = ops.AddFn.apply(y, addend)
    return ops.ActFn.apply(y, act) if act != ops.ACT_NONE else y
