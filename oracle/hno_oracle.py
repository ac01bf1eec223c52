"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A functional torch-CPU restatement of the reference's spectral-operator hot path.
It is the *checker* for the HIP kernels, never the product path: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status: the reference ships no tests or golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned by golden vectors generated in the
build container by importing the reference itself (``tests/golden/make_golden.py``)
and checked in ``tests/test_oracle_golden.py``.

Everything is written as plain functions over a ``state_dict`` (name -> tensor) so the
same weights can be fed to the reference modules, to this oracle and to the HIP-backed
modules.  Citations are ``file:line`` into the reference tree.

Two formulations of the Hartley transform are provided on purpose:
  * ``dhtn``          -- FFT based, op-for-op what nets/dht.py:16-36 does (used for the
                         CPU baseline timing, it issues the same ATen sequence);
  * ``cas_matrix`` /  ``dht_crop_dense`` / ``pad_idht_dense`` -- the pruned separable
                         formulation the HIP kernels implement, in dense matrix form
                         (fp64 capable), used to cross-check both.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946


# --------------------------------------------------------------------------------------
# Discrete Hartley transform (nets/dht.py:16-66)
# --------------------------------------------------------------------------------------
def dhtn(x, dims, inverse=False):
    """H(x) = Re F(x) - Im F(x); forward carries 1/prod(N), the 'inverse' is the same
    forward-sign transform left unscaled (nets/dht.py:29-34)."""
    spec = torch.fft.fftn(x, dim=dims, norm='backward' if inverse else 'forward')
    return spec.real - spec.imag


def dht2(x, inverse=False):
    return dhtn(x, (-2, -1), inverse)


def dht3(x, inverse=False):
    return dhtn(x, (-3, -2, -1), inverse)


def clamp_modes(modes, spatial):
    """Shared-weight mode clamping: 2m > s  ->  m = s // 2 (nets/hnosegxs.py:382-387,
    nets/hartley_operator.py:172-178)."""
    return tuple(s // 2 if 2 * m > s else m for m, s in zip(modes, spatial))


def kept_index(n, m):
    """Indices [0..m) U [n-m..n) -- the '[low | high]' corner order of
    nets/hnosegxs.py:393-410."""
    return torch.cat([torch.arange(0, m), torch.arange(n - m, n)])


def signed_freq(n, m):
    """Signed frequencies of kept_index(n, m): 0..m-1, -m..-1."""
    return torch.cat([torch.arange(0, m), torch.arange(-m, 0)])


def crop_modes(spec, modes):
    """Gather the kept corner block along the trailing len(modes) axes."""
    nd = len(modes)
    for ax, m in enumerate(modes):
        dim = spec.ndim - nd + ax
        spec = spec.index_select(dim, kept_index(spec.shape[dim], m))
    return spec


def pad_modes(z, spatial):
    """Scatter a [low | high] block into a zero array of size `spatial`
    (nets/hnosegxs.py:459-490)."""
    nd = len(spatial)
    out = z
    for ax, s in enumerate(spatial):
        dim = z.ndim - nd + ax
        m = out.shape[dim] // 2
        assert s >= 2 * m
        lo = out.narrow(dim, 0, m)
        hi = out.narrow(dim, out.shape[dim] - m, m)
        shape = list(out.shape)
        shape[dim] = s - 2 * m
        out = torch.cat([lo, out.new_zeros(shape), hi], dim=dim)
    return out


def transform_crop(x, modes):
    """nets/hnosegxs.py:332-410 (TransformCrop)."""
    nd = len(modes)
    modes = clamp_modes(modes, x.shape[-nd:])
    return crop_modes(dhtn(x, tuple(range(-nd, 0))), modes)


def pad_inverse(z, spatial):
    """nets/hnosegxs.py:413-494 (PadInverse)."""
    nd = len(spatial)
    return dhtn(pad_modes(z, spatial), tuple(range(-nd, 0)), inverse=True)


# ---- dense (pruned separable) formulation -------------------------------------------
def cas_matrix(n, rows=None, dtype=torch.float64):
    """C[k, j] = cos(2 pi k j / n) + sin(2 pi k j / n) for k in rows."""
    k = torch.arange(n) if rows is None else rows
    ang = (2.0 * math.pi / n) * (k.to(torch.float64)[:, None] * torch.arange(n, dtype=torch.float64)[None, :])
    return (torch.cos(ang) + torch.sin(ang)).to(dtype)


def _dft_rows(n, m):
    k = signed_freq(n, m).to(torch.float64)
    ang = (2.0 * math.pi / n) * (k[:, None] * torch.arange(n, dtype=torch.float64)[None, :])
    return torch.complex(torch.cos(ang), -torch.sin(ang))  # e^{-i theta k j}


def dht_crop_dense(x, modes, scale=None):
    """TransformCrop as a pruned separable DFT (only the kept rows of each DFT matrix),
    then Re - Im.  `scale` defaults to 1/prod(N)."""
    nd = len(modes)
    spatial = x.shape[-nd:]
    modes = clamp_modes(modes, spatial)
    acc = x.to(torch.complex128)
    for ax, (n, m) in enumerate(zip(spatial, modes)):
        dim = x.ndim - nd + ax
        acc = torch.movedim(torch.tensordot(acc, _dft_rows(n, m), dims=([dim], [1])), -1, dim)
    if scale is None:
        scale = 1.0 / float(np.prod(spatial))
    return ((acc.real - acc.imag) * scale).to(x.dtype)


def pad_idht_dense(z, spatial, scale=1.0):
    """PadInverse as sum_k z[k] cas(phi(k, n)) over kept modes only."""
    nd = len(spatial)
    acc = z.to(torch.complex128) * (1.0 - 1.0j)
    for ax, n in enumerate(spatial):
        dim = z.ndim - nd + ax
        m = z.shape[dim] // 2
        acc = torch.movedim(torch.tensordot(acc, _dft_rows(n, m).conj(), dims=([dim], [0])), -1, dim)
    return (acc.real * scale).to(z.dtype)


# --------------------------------------------------------------------------------------
# Spectral operators
# --------------------------------------------------------------------------------------
def reverse_freq(x, dims):
    """x[N-k]: flip then roll by one (nets/hartley_operator.py:320-333)."""
    return torch.roll(torch.flip(x, dims), [1] * len(dims), dims)


def hartley_mix(weight, z, weights_type='shared'):
    """Frequency-domain channel mix on an already cropped spectrum
    (nets/hartley_operator.py:287-299, 302-317)."""
    nd = z.ndim - 2
    letters = 'dhw'[-nd:]
    if weights_type == 'shared':
        return torch.einsum(f'oi,bi{letters}->bo{letters}', weight, z)
    dims = list(range(-nd, 0))
    zr, wr = reverse_freq(z, dims), reverse_freq(weight, dims)
    eq = f'oi{letters},bi{letters}->bo{letters}'
    return 0.5 * (torch.einsum(eq, weight, z + zr) + torch.einsum(eq, wr, z - zr))


def hartley_operator(x, weight, modes=None, bias=None, weights_type='shared', use_transform=True):
    """nets/hartley_operator.py:90-299.  With use_transform the mix is applied on the 2^nd
    corners of the full spectrum (individual weights: reverse taken on the FULL spectrum,
    :198-200), bias is added to the padded spectrum, SELU is applied in the frequency
    domain (:262-267), then the unscaled transform (:269)."""
    nd = x.ndim - 2
    if not use_transform:
        y = hartley_mix(weight, x, weights_type)
        return y if bias is None else y + bias
    spatial = x.shape[-nd:]
    dims = tuple(range(-nd, 0))
    if weights_type == 'shared':
        modes = clamp_modes(modes, spatial)
    else:
        assert all(s >= 2 * m for s, m in zip(spatial, modes))
    spec = dhtn(x, dims)
    letters = 'dhw'[-nd:]
    if weights_type == 'shared':
        mixed = torch.einsum(f'oi,bi{letters}->bo{letters}', weight, crop_modes(spec, modes))
    else:
        spec_r = crop_modes(reverse_freq(spec, list(dims)), modes)
        w_r = reverse_freq(weight, list(dims))  # weight grid is exactly the 2m block
        z = crop_modes(spec, modes)
        eq = f'oi{letters},bi{letters}->bo{letters}'
        mixed = 0.5 * (torch.einsum(eq, weight, z + spec_r) + torch.einsum(eq, w_r, z - spec_r))
    full = pad_modes(mixed, spatial)
    if bias is not None:
        full = full + bias
    return dhtn(F.selu(full), dims, inverse=True)


def fourier_operator(x, weight_real, weight_imag, modes, bias=None, weights_type='shared'):
    """nets/fourier_operator.py:148-211 (3-D) / :108-146 (2-D): rfftn(norm=forward),
    complex mix on the low/high corners of all but the last axis and the low corner of
    the last, zero pad, irfftn(norm=forward) (i.e. unscaled inverse)."""
    nd = x.ndim - 2
    spatial = x.shape[-nd:]
    dims = tuple(range(-nd, 0))
    if weights_type == 'shared':
        modes = clamp_modes(modes, spatial)
    else:
        assert all(s >= 2 * m for s, m in zip(spatial, modes))
    spec = torch.fft.rfftn(x, dim=dims, norm='forward')
    w = torch.complex(weight_real, weight_imag)
    lead = spec
    for ax, m in enumerate(modes[:-1]):
        dim = spec.ndim - nd + ax
        lead = lead.index_select(dim, kept_index(lead.shape[dim], m))
    lead = lead.narrow(-1, 0, modes[-1])
    letters = 'dhw'[-nd:]
    if weights_type == 'shared':
        mixed = torch.einsum(f'oi,bi{letters}->bo{letters}', w, lead)
    else:
        mixed = torch.einsum(f'oi{letters},bi{letters}->bo{letters}', w, lead)
    full = mixed
    for ax, s in enumerate(spatial[:-1]):
        dim = full.ndim - nd + ax
        m = modes[ax]
        shape = list(full.shape)
        shape[dim] = s - 2 * m
        full = torch.cat([full.narrow(dim, 0, m), full.new_zeros(shape), full.narrow(dim, m, m)], dim=dim)
    if bias is not None:
        full = full + bias
    size = tuple([-1] * (nd - 1) + [spatial[-1]])
    return torch.fft.irfftn(full, s=size, dim=dims, norm='forward')


# --------------------------------------------------------------------------------------
# Layer utilities (nets/nets_utils.py)
# --------------------------------------------------------------------------------------
def padcrop_amounts(shape, target):
    """Per-axis (lo, hi) pad and crop; the odd element goes to the high side
    (nets/nets_utils.py:77-97)."""
    pads, crops = [], []
    for s, t in zip(shape, target):
        d = t - s
        if d >= 0:
            pads.append((d // 2, d - d // 2))
            crops.append((0, 0))
        else:
            pads.append((0, 0))
            crops.append(((-d) // 2, (-d) - (-d) // 2))
    return pads, crops


def spatial_padcrop(x, target):
    """nets/nets_utils.py:22-57."""
    nd = len(target)
    pads, crops = padcrop_amounts(x.shape[-nd:], target)
    if any(p != (0, 0) for p in pads):
        flat = []
        for lo, hi in reversed(pads):
            flat += [lo, hi]
        x = F.pad(x, flat)
    for ax, (lo, hi) in enumerate(crops):
        if lo or hi:
            dim = x.ndim - nd + ax
            x = x.narrow(dim, lo, x.shape[dim] - lo - hi)
    return x


def conv_act(x, weight, bias=None, stride=1, act='selu'):
    """ConvNormAct without normalisation (the SNN/SELU case, nets/nets_utils.py:156-174):
    padding is 'same' for stride 1, kernel//2 otherwise."""
    nd = x.ndim - 2
    conv = F.conv3d if nd == 3 else F.conv2d
    k = weight.shape[-1]
    pad = 'same' if stride == 1 else k // 2
    y = conv(x, weight, bias, stride=stride, padding=pad)
    return _activate(y, act)


def _activate(y, act):
    if act is None:
        return y
    return getattr(F, act)(y)


def group_norm1(x, weight, bias, eps=1e-5):
    """nn.GroupNorm(1, C) (nets/nets_utils.py:170)."""
    return F.group_norm(x, 1, weight, bias, eps)


# --------------------------------------------------------------------------------------
# HNOSeg-XS (nets/hnosegxs.py)
# --------------------------------------------------------------------------------------
def hnoxs_block(sd, prefix, x, modes, n_convs, act='selu', use_block_concat=True, weights_type='shared'):
    """HNOXSBlock.forward, nets/hnosegxs.py:253-279 (SELU path, no normalisation)."""
    nd = x.ndim - 2
    if f'{prefix}.mapping_conv.op.weight' in sd:
        x = conv_act(x, sd[f'{prefix}.mapping_conv.op.weight'], sd[f'{prefix}.mapping_conv.op.bias'], act=act)
    skip = x
    spatial = x.shape[-nd:]
    z = transform_crop(x, modes)
    for j in range(n_convs):  # NeuralOperatorBlock.forward, :307-329
        w = sd[f'{prefix}.conv_blocks.{j}.op.weight']
        x1 = hartley_mix(w, z, weights_type)
        wb = sd.get(f'{prefix}.conv_blocks.{j}.conv_branch.weight')   # use_conv_branch (:293-294, :308, :315-316)
        if wb is not None:
            x1 = x1 + (F.conv3d if nd == 3 else F.conv2d)(z, wb)
        z = _activate(x1 + z, act)
    u = _activate(pad_inverse(z, spatial), act)
    if use_block_concat:
        return conv_act(torch.cat([u, skip], dim=1), sd[f'{prefix}.conv_concat.op.weight'],
                        sd[f'{prefix}.conv_concat.op.bias'], act=act)
    return u + skip


def hnosegxs_forward(sd, x, num_transform_blocks, num_modes, use_resize=True, use_unet_skip=True,
                     use_block_concat=True, use_deep_supervision=False, weights_type='shared', act='selu',
                     output_activation='softmax'):
    """HNOSegXS.forward, nets/hnosegxs.py:145-182."""
    nd = x.ndim - 2
    image_size = tuple(x.shape[-nd:])
    if np.isscalar(num_modes):
        num_modes = (num_modes,) * nd
    h = x
    if use_resize:
        h = conv_act(h, sd['conv_in.op.weight'], sd['conv_in.op.bias'], stride=2, act=act)
    h = conv_act(h, sd['conv1.op.weight'], sd['conv1.op.bias'], act=act)
    ds = [h] if use_deep_supervision else []
    nb = len(num_transform_blocks)
    enc = {}
    for i, n_convs in enumerate(num_transform_blocks):
        if use_unet_skip and i > nb // 2:
            h = torch.cat([h, enc[nb - 1 - i]], dim=1)
        h = hnoxs_block(sd, f'layers.{i}', h, num_modes, n_convs, act, use_block_concat, weights_type)
        if use_deep_supervision:
            ds.append(h)
        if use_unet_skip and i < nb // 2:
            enc[i] = h
    if ds:
        h = torch.cat(ds, dim=1)
    if use_resize:
        h = F.interpolate(h, size=image_size, mode='trilinear' if nd == 3 else 'bilinear')
    conv = F.conv3d if nd == 3 else F.conv2d
    h = spatial_padcrop(conv(h, sd['conv_out.weight']), image_size)
    if output_activation == 'softmax':
        return F.softmax(h, dim=1)
    return _activate(h, output_activation)


# --------------------------------------------------------------------------------------
# FNO / FNOSeg / HNOSeg (nets/architectures.py:255-429, 511-608)
# --------------------------------------------------------------------------------------
def neural_operator_block(sd, prefix, x, modes, transform_type, weights_type='shared', act='selu',
                          use_block_skip=True):
    """_TransBlock.forward (nets/architectures.py:521-548) for NeuralOperatorBlock (SELU: no normalisation)."""
    if transform_type == 'Hartley':
        x1 = hartley_operator(x, sd[f'{prefix}.op.weight'], modes, None, weights_type, True)
    else:
        x1 = fourier_operator(x, sd[f'{prefix}.op.weight_real'], sd[f'{prefix}.op.weight_imag'], modes, None, weights_type)
    y = x1
    if f'{prefix}.conv_branch.weight' in sd:
        conv = F.conv3d if x.ndim == 5 else F.conv2d
        y = y + conv(x, sd[f'{prefix}.conv_branch.weight'], sd.get(f'{prefix}.conv_branch.bias'))
    y = _activate(y, act)
    if use_block_skip:
        if f'{prefix}.conv_concat.op.weight' in sd:
            return conv_act(torch.cat([y, x], dim=1), sd[f'{prefix}.conv_concat.op.weight'],
                            sd[f'{prefix}.conv_concat.op.bias'], act=act)
        return y + x
    return y


def neural_operator_seg_forward(sd, x, num_transform_blocks, num_modes, transform_type, weights_type='shared',
                                use_resize=True, use_block_skip=True, act='selu', output_activation='softmax',
                                use_deep_supervision=False):
    """_TransSeg.forward (nets/architectures.py:325-353); deep supervision: conv_ds over the concat of conv1's and every
    block's output (:341-343)."""
    nd = x.ndim - 2
    image_size = tuple(x.shape[-nd:])
    if np.isscalar(num_modes):
        num_modes = (num_modes,) * nd
    h = x
    if use_resize:
        h = conv_act(h, sd['conv_in.op.weight'], sd['conv_in.op.bias'], stride=2, act=act)
    h = conv_act(h, sd['conv1.op.weight'], sd['conv1.op.bias'], act=act)
    tensors = [h]
    for i in range(num_transform_blocks):
        h = neural_operator_block(sd, f'layers.{i}', h, num_modes, transform_type, weights_type, act, use_block_skip)
        tensors.append(h)
    if use_deep_supervision:
        h = conv_act(torch.cat(tensors, dim=1), sd['conv_ds.op.weight'], sd['conv_ds.op.bias'], act=act)
    if use_resize:
        h = F.interpolate(h, size=image_size, mode='trilinear' if nd == 3 else 'bilinear')
    conv = F.conv3d if nd == 3 else F.conv2d
    h = spatial_padcrop(conv(h, sd['conv_out.weight']), image_size)
    return F.softmax(h, dim=1) if output_activation == 'softmax' else _activate(h, output_activation)



# --------------------------------------------------------------------------------------
# V-Net-DS (nets/architectures.py:26-252)
# --------------------------------------------------------------------------------------
def _cna(sd, prefix, x, stride=1, act='elu', transpose=False):
    """ConvNormAct / ConvTransposeNormAct with GroupNorm(1, C) (use_snn=False path, nets/nets_utils.py:136-211)."""
    w, b = sd[f'{prefix}.op.weight'], sd.get(f'{prefix}.op.bias')
    k = w.shape[-1]
    if transpose:
        y = F.conv_transpose3d(x, w, b, stride=2, padding=k // 2, output_padding=1)
    else:
        y = F.conv3d(x, w, b, stride=stride, padding='same' if stride == 1 else k // 2)
    if f'{prefix}.normalization.weight' in sd:
        y = F.group_norm(y, 1, sd[f'{prefix}.normalization.weight'], sd[f'{prefix}.normalization.bias'])
    return _activate(y, act)


def vnetds_forward(sd, x, num_blocks, right_leg_indexes=None, use_resize=True, use_residual=True, act='elu'):
    """VNetDS.forward / encode / decode (nets/architectures.py:184-252)."""
    image_size = tuple(x.shape[2:])
    legs_idx = right_leg_indexes if right_leg_indexes is not None else [0]
    nsec = len(num_blocks)
    h = x
    if use_resize:
        h = _cna(sd, 'conv_in', h, stride=2, act=act)
    enc, legs = {}, {}
    for i in range(nsec):
        j, tmp = 0, h
        for _ in range(num_blocks[i]):
            h = _cna(sd, f'encode_layers.{i}.{j}', h, act=act)
            j += 1
        if use_residual:
            h = h + _cna(sd, f'encode_layers.{i}.{j}', tmp, act=act)
            j += 1
        if i != nsec - 1:
            enc[i] = h
            h = _cna(sd, f'encode_layers.{i}.{j}', h, stride=2, act=act)
        elif i in legs_idx:
            legs[i] = h
    for i in reversed(range(nsec - 1)):
        h = _cna(sd, f'decode_layers.{i}.0', h, act=act, transpose=True)
        h = spatial_padcrop(h, tuple(enc[i].shape[2:]))
        h = torch.cat([h, enc[i]], dim=1)
        j, tmp = 1, h
        for _ in range(num_blocks[i]):
            h = _cna(sd, f'decode_layers.{i}.{j}', h, act=act)
            j += 1
        if use_residual:
            h = h + _cna(sd, f'decode_layers.{i}.{j}', tmp, act=act)
        if i in legs_idx:
            legs[i] = h
    if len(legs) == 1:
        h = legs[0]
    else:
        ref = tuple(legs[0].shape[2:])
        h = torch.cat([F.interpolate(t, ref) for t in legs.values()], dim=1)
        h = _cna(sd, 'conv_ds', h, act=act)
    if use_resize:
        h = F.interpolate(h, size=image_size, mode='trilinear')
    h = spatial_padcrop(F.conv3d(h, sd['conv_out.weight']), image_size)
    return F.softmax(h, dim=1)


# --------------------------------------------------------------------------------------
# Hartley multi-head attention (nets/hartley_mha.py:136-222, 473-524)
# --------------------------------------------------------------------------------------
def _group3d(x, patch):
    pd, ph, pw = patch
    b, z, c, d, h, w = x.shape
    x = x.reshape(b, z, c, d // pd, pd, h // ph, ph, w // pw, pw).permute(0, 1, 2, 4, 6, 8, 3, 5, 7)
    return x.reshape(b, z, c * pd * ph * pw, d // pd, h // ph, w // pw)


def _ungroup3d(x, c, patch):
    pd, ph, pw = patch
    b, z, _, nd_, nh, nw = x.shape
    x = x.reshape(b, z, c, pd, ph, pw, nd_, nh, nw).permute(0, 1, 2, 6, 3, 7, 4, 8, 5)
    return x.reshape(b, z, c, nd_ * pd, nh * ph, nw * pw)


def hartley_mha(x, wq, wk, wv, wo, modes, patch=None, att_act='selu', x_key=None, x_value=None):
    """HartleyMultiHeadAttention._call for 3-D inputs without biases.  x_key / x_value: the optional second
    and third inputs (key = value = second input when only two are given)."""
    spatial = x.shape[-3:]
    assert all(s >= 2 * m for s, m in zip(spatial, modes))
    dims = (-3, -2, -1)
    q_s = crop_modes(dhtn(x, dims), modes)
    k_s = q_s if x_key is None else crop_modes(dhtn(x_key, dims), modes)
    v_s = k_s if x_value is None else crop_modes(dhtn(x_value, dims), modes)
    q = torch.einsum('zoi,bidhw->bzodhw', wq, q_s)
    k = torch.einsum('zoi,bidhw->bzodhw', wk, k_s)
    v = torch.einsum('zoi,bidhw->bzodhw', wv, v_s)
    if patch is not None:
        q, k, v = _group3d(q, patch), _group3d(k, patch), _group3d(v, patch)
    fshape = q.shape[3:]
    q, k, v = (t.reshape(t.shape[0], t.shape[1], t.shape[2], -1) for t in (q, k, v))
    att = torch.einsum('bzcq,bzck->bzqk', q, k) / math.sqrt(k.shape[2])
    att = _activate(att, att_act)
    out = torch.einsum('bzqk,bzck->bzcq', att, v)
    out = out.reshape(out.shape[0], out.shape[1], out.shape[2], *fshape)
    if patch is not None:
        out = _ungroup3d(out, wv.shape[1], patch)
    out = out.reshape(out.shape[0], out.shape[1] * out.shape[2], *out.shape[3:])
    out = torch.einsum('oi,bidhw->bodhw', wo, out)
    return pad_inverse(out, spatial)


# --------------------------------------------------------------------------------------
# Losses (nets/custom_losses.py) and label handling (experiments/utils.py:74-119)
# --------------------------------------------------------------------------------------
def corrcoef(y_pred, y_true):
    """nets/custom_losses.py:17-41; eps 1e-7 inside the sqrt."""
    axes = tuple(range(2, y_true.ndim))
    t = y_true - y_true.mean(dim=axes, keepdim=True)
    p = y_pred - y_pred.mean(dim=axes, keepdim=True)
    return (t * p).sum(axes) / torch.sqrt((t * t).sum(axes) * (p * p).sum(axes) + 1e-7)


def pcc_loss(y_pred, y_true):
    """nets/custom_losses.py:56-70."""
    return (1.0 - (corrcoef(y_pred, y_true) + 1.0) * 0.5).mean()


def dice_coef(y_pred, y_true):
    """nets/custom_losses.py:73-90."""
    axes = tuple(range(2, y_true.ndim))
    return 2.0 * (y_true * y_pred).sum(axes) / ((y_true + y_pred).sum(axes) + 1e-7)


def dice_loss(y_pred, y_true):
    return (1.0 - dice_coef(y_pred, y_true)).mean()


def exp_dice_loss(y_pred, y_true, exp=0.3):
    """nets/custom_losses.py:120-133."""
    d = dice_coef(y_pred, y_true).clamp(1e-7, 1.0 - 1e-7)
    return (-torch.log(d)).pow(exp).mean()


def to_categorical(y, num_classes=None):
    """experiments/utils.py:74-97: (B,1,...) labels -> (B,K,...) one-hot fp32."""
    assert y.shape[1] == 1
    lab = y[:, 0].to(torch.int64)
    if not num_classes:
        num_classes = int(lab.max()) + 1
    return torch.movedim(F.one_hot(lab, num_classes).to(torch.float32), -1, 1)


def remap_labels(label, mapping):
    """experiments/utils.py:100-119: every key is matched against the ORIGINAL labels."""
    out = label.clone()
    for old, new in mapping.items():
        out[label == old] = new
    return out


# --------------------------------------------------------------------------------------
# Input pipeline (experiments/utils.py:25-71, experiments/data_io/dataset.py:192-244)
# --------------------------------------------------------------------------------------
def normalize_modalities(data, mask_val=None, clip_val=None):
    """utils.py:25-71: per modality (leading axis): clip, mean / population std over voxels != mask_val,
    z-score, masked voxels -> 0.  float64 accumulation; pinned by golden G11 (reference output)."""
    data = np.asarray(data, dtype=np.float32)
    out = np.empty_like(data)
    for c in range(data.shape[0]):
        v = data[c]
        if clip_val is not None:
            v = np.clip(v, clip_val[0], clip_val[1])
        keep = np.ones(v.shape, dtype=bool) if mask_val is None else (v != mask_val)
        sel = v[keep].astype(np.float64)
        mean = np.float32(sel.mean())
        std = np.float32(np.sqrt(((sel - sel.mean()) ** 2).mean()))
        out[c] = np.where(keep, (v - mean) / std, np.float32(0))
    return out


def centre_affine(matrix, spatial):
    """dataset.py:192-199 + :218-221: homogeneous (x, y[, z]) matrix conjugated about size / 2 + 0.5 ->
    3 x 4 rows [M | t] in (x, y, z) voxel indices (2-D: identity z row)."""
    matrix = np.asarray(matrix, dtype=np.float64)
    n = matrix.shape[0]
    off = np.asarray(tuple(spatial)[::-1], dtype=np.float64) / 2.0 + 0.5
    a, b = np.eye(n), np.eye(n)
    a[:-1, -1], b[:-1, -1] = off, -off
    full = a @ matrix @ b
    if n == 3:
        m = np.eye(4)
        m[:2, :2], m[:2, 3] = full[:2, :2], full[:2, 2]
        full = m
    return full[:3, :4]


def affine_nearest(x, m12, cval=0.0, flips=()):
    """dataset.py:202-244 through ITK's published semantics (SimpleITK is NOT installed in this image, so this
    function is NOT pinned by a reference run -- "parity unpinned"): ResampleImageFilter with an AffineTransform
    (centre 0), unit spacing, zero origin, nearest-neighbour interpolation: the continuous input index of output
    voxel p is M p + t; it is inside the buffer iff -0.5 <= c < size - 0.5 per axis; the nearest index is
    floor(c + 0.5); outside voxels take `cval`.  `flips`: array axes (1-based spatial, axis 0 = channel) reversed
    AFTER resampling.  x: (C, D, H, W) or (C, H, W) numpy."""
    x = np.asarray(x)
    sp = x.shape[1:]
    D, H, W = ((1,) + tuple(sp)) if len(sp) == 2 else sp
    x4 = x.reshape((x.shape[0], D, H, W))
    z, y, xx = np.meshgrid(np.arange(D, dtype=np.float64), np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64),
                           indexing='ij')
    m = np.asarray(m12, dtype=np.float64)
    cx = m[0, 0] * xx + m[0, 1] * y + m[0, 2] * z + m[0, 3]
    cy = m[1, 0] * xx + m[1, 1] * y + m[1, 2] * z + m[1, 3]
    cz = m[2, 0] * xx + m[2, 1] * y + m[2, 2] * z + m[2, 3]
    inside = (cx >= -0.5) & (cx < W - 0.5) & (cy >= -0.5) & (cy < H - 0.5) & (cz >= -0.5) & (cz < D - 0.5)
    ix = np.where(inside, np.floor(cx + 0.5), 0).astype(np.int64)
    iy = np.where(inside, np.floor(cy + 0.5), 0).astype(np.int64)
    iz = np.where(inside, np.floor(cz + 0.5), 0).astype(np.int64)
    out = np.where(inside[None], x4[:, iz, iy, ix], np.asarray(cval, dtype=x.dtype)).reshape(x.shape)
    for ax in flips:
        out = np.flip(out, ax)
    return np.ascontiguousarray(out)


# --------------------------------------------------------------------------------------
# Convenience: one fwd + loss + bwd step of HNOSeg-XS on CPU (the cpu_baseline "port")
# --------------------------------------------------------------------------------------
def hnosegxs_step(sd, x, labels, num_transform_blocks, num_modes, loss='pcc'):
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    y = hnosegxs_forward(params, x, num_transform_blocks, num_modes)
    onehot = to_categorical(labels, y.shape[1])
    val = {'pcc': pcc_loss, 'dice': dice_loss, 'expdice': exp_dice_loss}[loss](y, onehot)
    val.backward()
    return y.detach(), val.detach(), {k: p.grad for k, p in params.items()}
