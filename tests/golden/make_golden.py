#!/usr/bin/env python3
"""Generate golden fixtures by importing the REFERENCE (build container only).

    python tests/golden/make_golden.py [/root/reference [fixture function names...]]

The reference tree never travels to the GPU box; only the .npz files written here do.
Inputs come from the closed-form generators in _inputs.py, so fixtures hold weights and
expected outputs only.  Re-running this script must reproduce the committed files
(up to FFT-library rounding).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _inputs import formula_tensor, formula_labels, sample_indices, CROP_CASES, SMALL_MODELS, NOSEG_MODELS  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
sys.path.insert(0, REF)
import nets  # noqa: E402  (the reference package)
from nets import custom_losses, dht, hnosegxs, nets_utils  # noqa: E402
from nets.hartley_operator import HartleyOperator  # noqa: E402
from nets.fourier_operator import FourierOperator  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(max(1, os.cpu_count() or 1))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays')


def grads_of(out, cot, wrt):
    gs = torch.autograd.grad((out * cot).sum(), wrt, allow_unused=True)
    return [None if g is None else g.detach().numpy() for g in gs]


# ---------------------------------------------------------------------------- G1: dhtn
def g1_dht():
    out = {}
    for tag, shape in enumerate([(2, 3, 13, 15, 11), (1, 2, 33, 33, 33)]):
        for dt in (np.float32, np.float64):
            x = T(formula_tensor(shape, tag, dt))
            key = f's{tag}_{np.dtype(dt).name}'
            f3 = dht.dht3(x)
            i3 = dht.dht3(x, is_inverse=True)
            f2 = dht.dht2(x)
            i2 = dht.dht2(x, is_inverse=True)
            if tag == 0:
                out[f'{key}_fwd3'], out[f'{key}_inv3'] = f3.numpy(), i3.numpy()
                out[f'{key}_fwd2'], out[f'{key}_inv2'] = f2.numpy(), i2.numpy()
            else:
                idx = sample_indices(x.numel(), 4096, tag)
                out[f'{key}_idx'] = idx
                out[f'{key}_fwd3'], out[f'{key}_inv3'] = f3.numpy().ravel()[idx], i3.numpy().ravel()[idx]
                out[f'{key}_fwd2'] = f2.numpy().ravel()[idx]
            rt = dht.dht3(dht.dht3(x), is_inverse=True)
            out[f'{key}_roundtrip_err'] = np.array(float((rt - x).abs().max()))
    save('g1_dht.npz', **out)


# ------------------------------------------------------- G2: TransformCrop / PadInverse


def g2_crop_pad():
    out = {}
    for ci, (b, c, sp, modes) in enumerate(CROP_CASES):
        tc = hnosegxs.TransformCrop(modes, 5)
        pi = hnosegxs.PadInverse(5)
        x = T(formula_tensor((b, c) + sp, 10 + ci)).requires_grad_(True)
        z = tc(x)
        cot_z = T(formula_tensor(tuple(z.shape), 20 + ci))
        (gx,) = grads_of(z, cot_z, [x])
        zin = T(formula_tensor(tuple(z.shape), 30 + ci)).requires_grad_(True)
        y = pi(zin, sp)
        cot_y = T(formula_tensor(tuple(y.shape), 40 + ci))
        (gz,) = grads_of(y, cot_y, [zin])
        k = f'c{ci}'
        out[f'{k}_zshape'] = np.array(z.shape)
        out[f'{k}_crop'] = z.detach().numpy()
        out[f'{k}_crop_gradx_idx'] = idx = sample_indices(x.numel(), 4096, ci)
        out[f'{k}_crop_gradx'] = gx.ravel()[idx]
        out[f'{k}_pad_idx'] = idy = sample_indices(y.numel(), 4096, ci + 1)
        out[f'{k}_pad'] = y.detach().numpy().ravel()[idy]
        out[f'{k}_pad_gradz'] = gz
    save('g2_crop_pad.npz', **out)


# ------------------------------------------------ G3: HartleyOperator / FourierOperator
def g3_operators():
    out = {}
    ci_, co_, sp, modes = 3, 4, (12, 10, 14), (3, 2, 4)
    x_np = formula_tensor((2, ci_) + sp, 50)
    case = 0
    for cls, name in ((HartleyOperator, 'hartley'), (FourierOperator, 'fourier')):
        for wt in ('shared', 'individual'):
            for use_transform in (True, False):
                for use_bias in (False, True):
                    torch.manual_seed(100 + case)
                    op = cls(ci_, co_, modes, use_bias=use_bias, weights_type=wt, use_transform=use_transform)
                    if use_bias:
                        with torch.no_grad():
                            op.bias.copy_(T(formula_tensor(tuple(op.bias.shape), 60 + case)) * 0.1)
                    key = f'{name}_{wt}_t{int(use_transform)}_b{int(use_bias)}'
                    if use_transform:
                        x = T(x_np).requires_grad_(True)
                    elif name == 'hartley':
                        x = T(formula_tensor((2, ci_) + tuple(2 * m for m in modes), 70 + case)).requires_grad_(True)
                    else:  # Fourier, already in the (complex) frequency domain
                        shp = (2, ci_, 2 * modes[0], 2 * modes[1], modes[2])
                        x = torch.complex(T(formula_tensor(shp, 70 + case)), T(formula_tensor(shp, 170 + case)))
                        x = x.requires_grad_(True)
                    y = op(x)
                    params = dict(op.named_parameters())
                    if y.is_complex():
                        cot = torch.complex(T(formula_tensor(tuple(y.shape), 80 + case)),
                                            T(formula_tensor(tuple(y.shape), 180 + case)))
                        loss = (y * cot.conj()).real.sum()
                        gs = torch.autograd.grad(loss, [x] + list(params.values()))
                        gs = [g.detach().numpy() for g in gs]
                    else:
                        cot = T(formula_tensor(tuple(y.shape), 80 + case))
                        gs = grads_of(y, cot, [x] + list(params.values()))
                    out[f'{key}_case'] = np.array(case)
                    out[f'{key}_y'] = y.detach().numpy()
                    out[f'{key}_gx'] = gs[0]
                    for (pn, p), g in zip(params.items(), gs[1:]):
                        out[f'{key}_p_{pn}'] = p.detach().numpy()
                        out[f'{key}_g_{pn}'] = g
                    case += 1
    save('g3_operators.npz', **out)


# ---------------------------------------------------------------------- G5: losses
def g5_losses():
    out = {}
    shape = (2, 4, 9, 10, 11)
    logits = T(formula_tensor(shape, 90))
    yp = torch.softmax(logits, dim=1).requires_grad_(True)
    lab = formula_labels((2, 1, 9, 10, 11), 4, 3)
    lab[1][lab[1] == 2] = 1  # sample 1 has an all-zero channel (label 2 absent): eps path
    onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), 4).float(), -1, 1)
    out['labels'] = lab
    out['corrcoef'] = custom_losses.corrcoef(yp, onehot).detach().numpy()
    out['dice_coef'] = custom_losses.dice_coef(yp, onehot).detach().numpy()
    for name, fn in (('pcc', custom_losses.PCCLoss()), ('dice', custom_losses.DiceLoss()),
                     ('expdice', custom_losses.ExpDiceLoss(0.3))):
        val = fn(yp, onehot)
        (g,) = torch.autograd.grad(val, [yp])
        out[f'{name}_loss'] = val.detach().numpy()
        out[f'{name}_grad'] = g.numpy()
    save('g5_losses.npz', **out)


# ----------------------------------------------------------- G6: full HNOSeg-XS (cfg1)
def g6_hnosegxs():
    torch.manual_seed(0)
    model = nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14))
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params == 28248, n_params  # README.md:57-63 self check
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    out = {f'sd::{k}': v for k, v in sd.items()}
    out['n_params'] = np.array(n_params)
    for tag, shape in (('64', (1, 4, 64, 64, 64)), ('odd', (2, 4, 40, 36, 44))):
        x = T(formula_tensor(shape, 7))
        lab = formula_labels((shape[0], 1) + shape[2:], 4, 5)
        onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), 4).float(), -1, 1)
        model.zero_grad()
        y = model(x)
        loss = custom_losses.PCCLoss()(y, onehot)
        loss.backward()
        idx = sample_indices(y.numel(), 4096, 2)
        out[f'{tag}_shape'] = np.array(shape)
        out[f'{tag}_y_idx'] = idx
        out[f'{tag}_y'] = y.detach().numpy().ravel()[idx]
        out[f'{tag}_y_sum'] = np.array(y.detach().double().sum().item())
        out[f'{tag}_loss'] = loss.detach().numpy()
        for k, p in model.named_parameters():
            out[f'{tag}_grad::{k}'] = p.grad.detach().numpy().copy()
        # the same step with the REFERENCE model in float64: the rounding-free truth, used to show
        # that fp32 round-off (reference and HIP alike) -- not the algorithm -- sets the error floor
        import copy
        m64 = copy.deepcopy(model).double()
        m64.zero_grad()
        y64 = m64(x.double())
        loss64 = custom_losses.PCCLoss()(y64, onehot.double())
        loss64.backward()
        out[f'{tag}_y64'] = y64.detach().numpy().ravel()[idx].astype(np.float32)
        out[f'{tag}_loss64'] = loss64.detach().numpy()
        for k, p in m64.named_parameters():
            out[f'{tag}_grad64::{k}'] = p.grad.detach().numpy().astype(np.float32)
    save('g6_hnosegxs.npz', **out)


# ------------------------------- G6-128: HNOSeg-XS at the metric's own size (cfg2 grid, 128^3 -> 65^3)
def g6_128():
    """the reference HNOSeg-XS (same seed-0 weights as g6_hnosegxs.npz, not stored again) on one
    (1, 4, 128^3) formula volume: 4 096 sampled outputs, output sum, loss, all 28 248 gradients, in fp32 and with the
    reference run in float64.  A batch of 2 is the same volume stacked (PCC is a mean over (b, c): same loss, same grads)."""
    import copy
    torch.manual_seed(0)
    model = nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14))
    out = {}
    shape = (1, 4, 128, 128, 128)
    x = T(formula_tensor(shape, 7))
    lab = formula_labels((1, 1) + shape[2:], 4, 5)
    onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), 4).float(), -1, 1)
    model.zero_grad()
    y = model(x)
    loss = custom_losses.PCCLoss()(y, onehot)
    loss.backward()
    idx = sample_indices(y.numel(), 4096, 2)
    out['shape'] = np.array(shape)
    out['y_idx'] = idx
    out['y'] = y.detach().numpy().ravel()[idx]
    out['y_sum'] = np.array(y.detach().double().sum().item())
    out['y_chan_sum'] = y.detach().double().sum(dim=(0, 2, 3, 4)).numpy()
    out['loss'] = loss.detach().numpy()
    for k, p in model.named_parameters():
        out[f'grad::{k}'] = p.grad.detach().numpy().copy()
    del y, loss
    m64 = copy.deepcopy(model).double()
    m64.zero_grad()
    y64 = m64(x.double())
    loss64 = custom_losses.PCCLoss()(y64, onehot.double())
    loss64.backward()
    out['y64'] = y64.detach().numpy().ravel()[idx].astype(np.float32)
    out['loss64'] = loss64.detach().numpy()
    for k, p in m64.named_parameters():
        out[f'grad64::{k}'] = p.grad.detach().numpy().astype(np.float32)
    save('g6_128.npz', **out)


# ---------------------------- G6s: small, well-conditioned HNOSeg-XS variants (strict 1e-4)
def g6s_small_models():
    from _inputs import formula_volume
    out = {}
    for name, (kw, shape) in SMALL_MODELS.items():
        torch.manual_seed(11)
        model = nets.HNOSegXS(**kw)
        for k, v in model.state_dict().items():
            out[f'{name}::sd::{k}'] = v.detach().numpy().copy()
        K = kw['out_channels']
        x = T(formula_volume(shape, 3))
        lab = formula_labels((shape[0], 1) + shape[2:], K, 2)
        onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), K).float(), -1, 1)
        for lname, fn in (('pcc', custom_losses.PCCLoss()), ('dice', custom_losses.DiceLoss())):
            model.zero_grad()
            y = model(x)
            loss = fn(y, onehot)
            loss.backward()
            out[f'{name}::{lname}::loss'] = loss.detach().numpy()
            for k, p in model.named_parameters():
                out[f'{name}::{lname}::grad::{k}'] = p.grad.detach().numpy().copy()
        out[f'{name}::y'] = y.detach().numpy()
    save('g6s_small_models.npz', **out)


# ------------------------------ G6b: one HNOXSBlock with use_conv_branch=True (nets/hnosegxs.py:185-329)
def g6b_xsblock_branch():
    from _inputs import XSBLOCK_BRANCH as cfg
    torch.manual_seed(5)
    blk = hnosegxs.HNOXSBlock(cfg['num_convs'], cfg['in_channels'], cfg['out_channels'], cfg['num_modes'],
                              use_conv_branch=True)
    blk.apply(nets_utils.init_weights_for_snn)
    out = {f'sd::{k}': v.detach().numpy().copy() for k, v in blk.state_dict().items()}
    x = T(formula_volume_small(cfg['shape'])).requires_grad_(True)
    y = blk(x)
    cot = T(formula_tensor(tuple(y.shape), 61))
    gs = torch.autograd.grad((y * cot).sum(), [x] + list(blk.parameters()))
    out['y'] = y.detach().numpy()
    out['gx'] = gs[0].numpy()
    for (k, _), g in zip(blk.named_parameters(), gs[1:]):
        out[f'grad::{k}'] = g.numpy()
    save('g6b_xsblock_branch.npz', **out)


def formula_volume_small(shape):
    from _inputs import formula_volume
    return formula_volume(shape, 9)


# ------------------------------------------ G10: 2-D (ndim = 4) crop / pad and operators
def g10_two_d():
    from _inputs import CROP_CASES_2D
    out = {}
    for ci, (b, c, sp, modes) in enumerate(CROP_CASES_2D):
        tc = hnosegxs.TransformCrop(modes, 4)
        pi = hnosegxs.PadInverse(4)
        x = T(formula_tensor((b, c) + sp, 210 + ci)).requires_grad_(True)
        z = tc(x)
        (gx,) = grads_of(z, T(formula_tensor(tuple(z.shape), 220 + ci)), [x])
        zin = T(formula_tensor(tuple(z.shape), 230 + ci)).requires_grad_(True)
        y = pi(zin, sp)
        (gz,) = grads_of(y, T(formula_tensor(tuple(y.shape), 240 + ci)), [zin])
        k = f'c{ci}'
        out[f'{k}_crop'], out[f'{k}_crop_gradx'] = z.detach().numpy(), gx
        out[f'{k}_pad'], out[f'{k}_pad_gradz'] = y.detach().numpy(), gz
    ci_, co_, sp, modes = 3, 4, (12, 15), (3, 4)
    x_np = formula_tensor((2, ci_) + sp, 250)
    case = 0
    for cls, name in ((HartleyOperator, 'hartley'), (FourierOperator, 'fourier')):
        for wt in ('shared', 'individual'):
            for use_transform in (True, False):
                for use_bias in (False, True):
                    torch.manual_seed(300 + case)
                    op = cls(ci_, co_, modes, use_bias=use_bias, weights_type=wt, use_transform=use_transform, ndim=4)
                    if use_bias:
                        with torch.no_grad():
                            op.bias.copy_(T(formula_tensor(tuple(op.bias.shape), 260 + case)) * 0.1)
                    key = f'{name}_{wt}_t{int(use_transform)}_b{int(use_bias)}'
                    if use_transform:
                        x = T(x_np).requires_grad_(True)
                    elif name == 'hartley':
                        x = T(formula_tensor((2, ci_) + tuple(2 * m for m in modes), 270 + case)).requires_grad_(True)
                    else:
                        shp = (2, ci_, 2 * modes[0], modes[1])
                        x = torch.complex(T(formula_tensor(shp, 270 + case)), T(formula_tensor(shp, 370 + case)))
                        x = x.requires_grad_(True)
                    y = op(x)
                    params = dict(op.named_parameters())
                    if y.is_complex():
                        cot = torch.complex(T(formula_tensor(tuple(y.shape), 280 + case)),
                                            T(formula_tensor(tuple(y.shape), 380 + case)))
                        loss = (y * cot.conj()).real.sum()
                        gs = [g.detach().numpy() for g in torch.autograd.grad(loss, [x] + list(params.values()))]
                    else:
                        gs = grads_of(y, T(formula_tensor(tuple(y.shape), 280 + case)), [x] + list(params.values()))
                    out[f'{key}_case'] = np.array(case)
                    out[f'{key}_y'] = y.detach().numpy()
                    out[f'{key}_gx'] = gs[0]
                    for (pn, p_), g in zip(params.items(), gs[1:]):
                        out[f'{key}_p_{pn}'] = p_.detach().numpy()
                        out[f'{key}_g_{pn}'] = g
                    case += 1
    save('g10_two_d.npz', **out)


# ---------------------------- G11: input pipeline (normalisation values, augmentation random stream)
def g11_input():
    from _inputs import AUG_CASES, raw_modalities
    _stub_missing_modules()
    from experiments import utils as ref_utils
    from experiments.data_io import dataset as ref_ds
    out = {}
    # normalisation: smooth volume with a zero background (the mask) -- raw-MR-like offsets and scales
    vol = raw_modalities()
    out['norm_masked'] = ref_utils.normalize_modalities(vol, mask_val=0)
    out['norm_plain'] = ref_utils.normalize_modalities(vol)
    out['norm_clip'] = ref_utils.normalize_modalities(vol, mask_val=0, clip_val=(0, 600))
    # augmentation: the matrices ImageTransform hands to apply_transform and the flips it applies (SimpleITK itself is
    # not installed, so the resampling cannot be run here: apply_transform is replaced by a recorder)
    for name, (kw, shape) in AUG_CASES.items():
        rec = []
        ref_ds.apply_transform = lambda x, m, cval, rec=rec: (rec.append(np.array(m, dtype=np.float64)), x)[1]
        tr = ref_ds.ImageTransform(**kw)
        base = np.arange(int(np.prod(shape)), dtype=np.float32).reshape(shape)
        mats, outs, had = [], [], []
        for _ in range(12):
            n0 = len(rec)
            xo, yo = tr(base, base[:1])
            had.append(len(rec) > n0)
            mats.append(rec[n0] if len(rec) > n0 else np.eye(len(shape)))
            outs.append(np.ascontiguousarray(xo))       # shows the flips (identity resampling)
        out[f'{name}_had_matrix'] = np.array(had)
        out[f'{name}_matrices'] = np.stack(mats)
        out[f'{name}_flipped'] = np.stack(outs)
    save('g11_input.npz', **out)


# ------------------------------------------- G9: labels, padcrop, SNN init statistics
def g9_misc():
    sys.modules.setdefault('SimpleITK', type(sys)('SimpleITK'))
    ti = type(sys)('torchinfo')
    ti.summary = lambda *a, **k: 'summary'
    sys.modules.setdefault('torchinfo', ti)
    sys.modules.setdefault('torchview', type(sys)('torchview'))
    from experiments import utils as ref_utils
    out = {}
    lab = formula_labels((2, 1, 5, 6, 7), 5, 1)
    out['labels'] = lab
    out['onehot5'] = ref_utils.to_categorical(T(lab), 5).numpy()
    out['onehot_auto'] = ref_utils.to_categorical(T(lab)).numpy()
    mapping = {4: 3, 3: 1, 1: 2}
    out['remap_keys'] = np.array(list(mapping.keys()))
    out['remap_vals'] = np.array(list(mapping.values()))
    out['remapped'] = ref_utils.remap_labels(T(lab), mapping).numpy()
    x = T(formula_tensor((1, 2, 7, 8, 9), 3))
    targets = [(7, 8, 9), (10, 11, 12), (4, 5, 6), (9, 5, 9), (8, 8, 8)]
    out['padcrop_targets'] = np.array(targets)
    for i, t in enumerate(targets):
        out[f'padcrop_{i}'] = nets_utils.spatial_padcrop(x, list(t)).numpy()
    save('g9_misc.npz', **out)


# ---------------------------------------------- G7: tiny FNOSeg / HNOSeg variants (strict 1e-4)
def g7_noseg_models():
    from _inputs import formula_volume
    out = {}
    for name, (kw, shape) in NOSEG_MODELS.items():
        torch.manual_seed(21)
        model = nets.NeuralOperatorSeg(**kw)
        for k, v in model.state_dict().items():
            out[f'{name}::sd::{k}'] = v.detach().numpy().copy()
        K = kw['out_channels']
        x = T(formula_volume(shape, 4))
        lab = formula_labels((shape[0], 1) + shape[2:], K, 6)
        onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), K).float(), -1, 1)
        model.zero_grad()
        y = model(x)
        loss = custom_losses.PCCLoss()(y, onehot)
        loss.backward()
        out[f'{name}::y'] = y.detach().numpy()
        out[f'{name}::loss'] = loss.detach().numpy()
        for k, p in model.named_parameters():
            out[f'{name}::grad::{k}'] = p.grad.detach().numpy().copy()
    save('g7_noseg_models.npz', **out)


# --------------------------------------------------------- G4: Hartley multi-head attention
def g4_mha():
    from _inputs import MHA_CASES, MHASEG_MODEL, formula_volume
    from nets.hartley_mha import HartleyMultiHeadAttention
    out = {}
    shape = (1, 6, 12, 14, 12)
    for ci, (cin, kd, heads, modes, patch, nin) in enumerate(MHA_CASES):
        torch.manual_seed(300 + ci)
        op = HartleyMultiHeadAttention(cin, kd, heads, modes, patch)
        xs = [T(formula_tensor(shape, 90 + ci + 7 * j)).requires_grad_(True) for j in range(nin)]
        y = op(xs[0] if nin == 1 else xs)
        cot = T(formula_tensor(tuple(y.shape), 95 + ci))
        params = dict(op.named_parameters())
        gs = grads_of(y, cot, xs + list(params.values()))
        k = f'm{ci}'
        out[f'{k}_y'] = y.detach().numpy()
        for j in range(nin):
            out[f'{k}_gx{j}'] = gs[j]
        for (pn, p), g in zip(params.items(), gs[nin:]):
            out[f'{k}_p_{pn}'] = p.detach().numpy()
            out[f'{k}_g_{pn}'] = g
    kw, mshape = MHASEG_MODEL
    torch.manual_seed(31)
    model = nets.HartleyMHASeg(**kw)
    for k, v in model.state_dict().items():
        out[f'seg::sd::{k}'] = v.detach().numpy().copy()
    K = kw['out_channels']
    x = T(formula_volume(mshape, 8))
    lab = formula_labels((mshape[0], 1) + mshape[2:], K, 9)
    onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), K).float(), -1, 1)
    y = model(x)
    loss = custom_losses.PCCLoss()(y, onehot)
    loss.backward()
    out['seg::y'] = y.detach().numpy()
    out['seg::loss'] = loss.detach().numpy()
    for k, p in model.named_parameters():
        out[f'seg::grad::{k}'] = p.grad.detach().numpy().copy()
    save('g4_mha.npz', **out)


# ------------------------------------------- G12: HartleyMultiHeadAttention with biases (use_bias=True)
def g12_mha_bias():
    from _inputs import MHA_BIAS_CASES
    from nets.hartley_mha import HartleyMultiHeadAttention
    out = {}
    for ci, (cin, kd, heads, modes, patch, nin, use_transform) in enumerate(MHA_BIAS_CASES):
        torch.manual_seed(400 + ci)
        op = HartleyMultiHeadAttention(cin, kd, heads, modes, patch, use_bias=True, use_transform=use_transform)
        with torch.no_grad():
            for j, (pn, p_) in enumerate(op.named_parameters()):
                if pn.startswith('bias'):
                    p_.copy_(T(formula_tensor(tuple(p_.shape), 410 + ci + j)) * 0.2)
        shape = (1, cin, 12, 14, 12) if use_transform else (1, cin) + tuple(2 * m for m in modes)
        xs = [T(formula_tensor(shape, 420 + ci + 7 * j)).requires_grad_(True) for j in range(nin)]
        y = op(xs[0] if nin == 1 else xs)
        params = dict(op.named_parameters())
        gs = grads_of(y, T(formula_tensor(tuple(y.shape), 430 + ci)), xs + list(params.values()))
        k = f'm{ci}'
        out[f'{k}_y'] = y.detach().numpy()
        for j in range(nin):
            out[f'{k}_gx{j}'] = gs[j]
        for (pn, p_), g in zip(params.items(), gs[nin:]):
            out[f'{k}_p_{pn}'] = p_.detach().numpy()
            out[f'{k}_g_{pn}'] = g
    save('g12_mha_bias.npz', **out)


# ------------------------------------------------------------- G13: tiny 2-D (ndim = 4) models
def g13_models_2d():
    from _inputs import MODELS_2D
    out = {}
    for name, (cls, kw, shape) in MODELS_2D.items():
        torch.manual_seed(61)
        model = getattr(nets, cls)(**kw)
        for k, v in model.state_dict().items():
            out[f'{name}::sd::{k}'] = v.detach().numpy().copy()
        K = kw['out_channels']
        x = T(formula_tensor(shape, 14))
        lab = formula_labels((shape[0], 1) + shape[2:], K, 16)
        onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), K).float(), -1, 1)
        y = model(x)
        loss = custom_losses.PCCLoss()(y, onehot)
        loss.backward()
        out[f'{name}::y'] = y.detach().numpy()
        out[f'{name}::loss'] = loss.detach().numpy()
        for k, p_ in model.named_parameters():
            out[f'{name}::grad::{k}'] = p_.grad.detach().numpy().copy()
    save('g13_models_2d.npz', **out)


# ------------------------------------------------------------- G7v: tiny V-Net-DS variants
def g7v_vnet_models():
    from _inputs import VNET_MODELS, formula_volume
    out = {}
    for name, (kw, shape) in VNET_MODELS.items():
        torch.manual_seed(41)
        model = nets.VNetDS(**kw)
        for k, v in model.state_dict().items():
            out[f'{name}::sd::{k}'] = v.detach().numpy().copy()
        K = kw['out_channels']
        x = T(formula_volume(shape, 5))
        lab = formula_labels((shape[0], 1) + shape[2:], K, 7)
        onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), K).float(), -1, 1)
        y = model(x)
        loss = custom_losses.DiceLoss()(y, onehot)
        loss.backward()
        out[f'{name}::y'] = y.detach().numpy()
        out[f'{name}::loss'] = loss.detach().numpy()
        for k, p in model.named_parameters():
            out[f'{name}::grad::{k}'] = p.grad.detach().numpy().copy()
    save('g7v_vnet_models.npz', **out)


# ------------------------- G15: BASELINE cfg3 (FNOSeg, 24 Fourier blocks) at its REAL volume size, fp32 and under bf16 autocast
def g15_cfg3_full_size():
    """The reference's NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier') (config_fnoseg.ini:46-52) on one (1, 4, 128^3) formula
    volume: 4 096 sampled outputs, output sum, loss and all 71 184 parameter gradients, in fp32, under
    torch.autocast('cpu', bfloat16) (train_test.py:154-160) and with the reference run in float64.  Until round 5 the full-size cfg3 test compared the bf16 kernels with the
    repo's own fp32 kernels only; with this fixture it is anchored to the reference at the benchmark's own size.  A batch of 2 is the same
    volume stacked (PCC is a mean over (b, c))."""
    torch.manual_seed(0)
    model = nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier')
    out = {}
    for k, v in model.state_dict().items():
        out[f'sd::{k}'] = v.detach().numpy().copy()
    shape = (1, 4, 128, 128, 128)
    x = T(formula_tensor(shape, 7))
    lab = formula_labels((1, 1) + shape[2:], 4, 5)
    onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), 4).float(), -1, 1)
    idx = None
    for tag in ('f32', 'bf16'):
        model.zero_grad()
        if tag == 'bf16':
            with torch.autocast(device_type='cpu', dtype=torch.bfloat16):
                y = model(x)
                loss = custom_losses.PCCLoss()(y, onehot)
        else:
            y = model(x)
            loss = custom_losses.PCCLoss()(y, onehot)
        loss.backward()
        if idx is None:
            idx = sample_indices(y.numel(), 4096, 2)
            out['shape'], out['y_idx'] = np.array(shape), idx
        out[f'{tag}::y'] = y.detach().float().numpy().ravel()[idx]
        out[f'{tag}::y_sum'] = np.array(y.detach().double().sum().item())
        out[f'{tag}::loss'] = loss.detach().float().numpy()
        for k, p in model.named_parameters():
            out[f'{tag}::grad::{k}'] = p.grad.detach().float().numpy().copy()
        del y, loss
    # ... and with the reference run in float64: the truth both fp32 evaluations (the reference's and the kernels') are measured against
    import copy
    m64 = copy.deepcopy(model).double()
    m64.zero_grad()
    y64 = m64(x.double())
    loss64 = custom_losses.PCCLoss()(y64, onehot.double())
    loss64.backward()
    out['f64::y'] = y64.detach().numpy().ravel()[idx].astype(np.float32)
    out['f64::loss'] = loss64.detach().numpy()
    for k, p in m64.named_parameters():
        out[f'f64::grad::{k}'] = p.grad.detach().numpy().astype(np.float32)
    save('g15_cfg3_full_size.npz', **out)


# ------------------------- G16: BASELINE cfg4 (V-Net-DS, 22.5 M parameters) at its REAL volume size, fp32 / bf16 autocast / float64
def g16_cfg4_full_size():
    """The reference's VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0..4]) (config_vnet-ds.ini:46-51) on one (1, 4, 160, 192, 128)
    formula volume: 4 096 sampled outputs, output sum, loss, and of the gradients (22.5 M values: too many to store) per parameter tensor
    its L2 norm and 512 sampled elements -- in fp32, under torch.autocast('cpu', bfloat16) and with the reference run in float64.  The
    weights are the constructor's under torch.manual_seed(0) (this package's constructors draw the same values: RNG-identical init); the
    fixture holds per-tensor sums and 8 sampled values so that the test can prove it before it compares anything."""
    import copy
    torch.manual_seed(0)
    model = nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4])
    out = {}
    names = [k for k, _ in model.named_parameters()]
    out['param_names'] = np.array(names)
    out['param_sum'] = np.array([float(p.detach().double().sum()) for _, p in model.named_parameters()])
    out['param_head'] = np.stack([np.resize(p.detach().numpy().ravel()[:8], 8) for _, p in model.named_parameters()])
    shape = (1, 4, 160, 192, 128)
    x = T(formula_tensor(shape, 9))
    lab = formula_labels((1, 1) + shape[2:], 4, 3)
    onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), 4).float(), -1, 1)
    idx = None
    gidx = {k: sample_indices(p.numel(), 512, 5) for k, p in model.named_parameters()}

    def record(tag, m, y, loss):
        nonlocal idx
        if idx is None:
            idx = sample_indices(y.numel(), 4096, 2)
            out['shape'], out['y_idx'] = np.array(shape), idx
        out[f'{tag}::y'] = y.detach().float().numpy().ravel()[idx]
        out[f'{tag}::y_sum'] = np.array(y.detach().double().sum().item())
        out[f'{tag}::loss'] = np.array(float(loss.detach()))
        out[f'{tag}::grad_norm'] = np.array([float(p.grad.detach().double().norm()) for _, p in m.named_parameters()])
        out[f'{tag}::grad_samples'] = np.concatenate([p.grad.detach().float().numpy().ravel()[gidx[k]] for k, p in m.named_parameters()])
    out['grad_sample_counts'] = np.array([len(gidx[k]) for k in names])
    for tag in ('f32', 'bf16'):
        model.zero_grad()
        if tag == 'bf16':
            with torch.autocast(device_type='cpu', dtype=torch.bfloat16):
                y = model(x)
                loss = custom_losses.PCCLoss()(y, onehot)
        else:
            y = model(x)
            loss = custom_losses.PCCLoss()(y, onehot)
        loss.backward()
        record(tag, model, y, loss)
        del y, loss
    m64 = nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4])     # (the model keeps activations as attributes: no deepcopy)
    m64.load_state_dict(model.state_dict())
    m64 = m64.double()
    m64.zero_grad()
    y64 = m64(x.double())
    loss64 = custom_losses.PCCLoss()(y64, onehot.double())
    loss64.backward()
    record('f64', m64, y64, loss64)
    save('g16_cfg4_full_size.npz', **out)


# ------------------------- G7b: the reference under torch.autocast(bfloat16) (train_test.py:154-160), beside its fp32 run
def g7b_bf16_models():
    """What `use_autocast` does to each model family on the reference itself (CPU autocast, bfloat16: convolutions take bf16
    operands and return bf16, GroupNorm / FFT / complex einsum stay fp32): outputs, loss and all gradients, stored next to
    the fp32 run of the same model so that the tests can state their tolerance as a multiple of the reference's own bf16-vs-fp32
    distance."""
    from _inputs import BF16_MODELS, formula_volume
    out = {}
    for name, (cls, kw, shape) in BF16_MODELS.items():
        torch.manual_seed(43)
        model = getattr(nets, cls)(**kw)
        for k, v in model.state_dict().items():
            out[f'{name}::sd::{k}'] = v.detach().numpy().copy()
        K = kw['out_channels']
        x = T(formula_volume(shape, 6))
        lab = formula_labels((shape[0], 1) + shape[2:], K, 8)
        onehot = torch.movedim(torch.nn.functional.one_hot(T(lab)[:, 0].long(), K).float(), -1, 1)
        for tag in ('f32', 'bf16'):
            model.zero_grad()
            if tag == 'bf16':
                with torch.autocast(device_type='cpu', dtype=torch.bfloat16):
                    y = model(x)
                    loss = custom_losses.PCCLoss()(y, onehot)
            else:
                y = model(x)
                loss = custom_losses.PCCLoss()(y, onehot)
            loss.backward()
            out[f'{name}::{tag}::y'] = y.detach().float().numpy()
            out[f'{name}::{tag}::loss'] = loss.detach().float().numpy()
            for k, p in model.named_parameters():
                out[f'{name}::{tag}::grad::{k}'] = p.grad.detach().float().numpy().copy()
    save('g7b_bf16_models.npz', **out)


# ----------------------------------------------- G8: training-loop trajectory of the reference
def _stub_missing_modules():
    sys.modules.setdefault('SimpleITK', type(sys)('SimpleITK'))
    ti = type(sys)('torchinfo')
    ti.summary = lambda *a, **k: 'summary'
    sys.modules.setdefault('torchinfo', ti)
    sys.modules.setdefault('torchview', type(sys)('torchview'))
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        mp = type(sys)('matplotlib')
        mp.use = lambda *a, **k: None
        sys.modules['matplotlib'] = mp
        sys.modules['matplotlib.pyplot'] = type(sys)('matplotlib.pyplot')


def g8_training():
    import tempfile
    _stub_missing_modules()
    from experiments import train_test as ref_tt
    ref_tt.plot_losses = lambda *a, **k: None        # plotting is out of scope
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), '..'))
    from _inputs import TRAIN_CASE, make_train_input
    data = make_train_input()
    torch.manual_seed(5)
    model = nets.HNOSegXS(**TRAIN_CASE['model'])
    out = {f'sd0::{k}': v.detach().numpy().copy() for k, v in model.state_dict().items()}
    opt = torch.optim.Adamax(model.parameters(), lr=TRAIN_CASE['lr'])
    nb = data.get_train_num_batches()
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=nb * TRAIN_CASE['epochs'], eta_min=TRAIN_CASE['eta_min'])
    with tempfile.TemporaryDirectory() as d:
        ref_tt.training(model, data, d, custom_losses.PCCLoss(), opt, sched, label_mapping=TRAIN_CASE['mapping'],
                        num_epochs=TRAIN_CASE['epochs'], selection_epoch_portion=0.5, checkpoint_epoch=2,
                        is_print=False, device='cpu')
        tl, vl = ref_tt.get_losses_from_file(os.path.join(d, 'stdout.txt'))
        ck = torch.load(os.path.join(d, 'model', 'checkpoint.pt'), weights_only=False)
        out['files'] = np.array(sorted(os.listdir(os.path.join(d, 'model'))))
    out['train_loss'], out['valid_loss'] = np.array(tl), np.array(vl)
    out['checkpoint_keys'] = np.array(sorted(ck.keys()))
    out['checkpoint_epoch'] = np.array(ck['epoch'])
    out['best_epoch'] = np.array(-1 if ck['best_epoch'] is None else ck['best_epoch'])
    out['final_lr'] = np.array(opt.param_groups[0]['lr'])
    for k, v in model.state_dict().items():
        out[f'sd1::{k}'] = v.detach().numpy().copy()
    save('g8_training.npz', **out)


# --------------------------------------- G14: the reference's testing() protocol (train_test.py:332-426)
def g14_testing():
    """Class maps exactly as the reference's own ``testing()`` produces them (model.eval(), batch 1, host arg max, label
    remapping), captured from its ``save_output`` calls, plus the top-2 probability margin per voxel so that the comparison
    can leave out voxels that are numerically tied."""
    import tempfile
    _stub_missing_modules()
    from experiments import train_test as ref_tt
    from _inputs import TEST_CASE, make_test_input
    captured = {}
    ref_tt.save_output = lambda arr, lists, i, d, origin, suffix: captured.__setitem__((i, suffix), np.asarray(arr).copy())
    torch.cuda.max_memory_reserved = lambda *a, **k: 0        # the reference prints CUDA statistics; there is no GPU here
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    torch.manual_seed(21)
    model = nets.HNOSegXS(**TEST_CASE['model'])
    out = {f'sd::{k}': v.detach().numpy().copy() for k, v in model.state_dict().items()}
    with tempfile.TemporaryDirectory() as d:
        ref_tt.testing(model, make_test_input(), d, label_mapping=TEST_CASE['mapping'], is_print=False, device='cpu')
        out['files'] = np.array(sorted(os.listdir(d)))
    model.eval()
    for i, (x, y) in enumerate(make_test_input().get_test_flow()):
        out[f'pred_{i}'] = captured[(i, '_pred')].astype(np.uint8)
        out[f'true_{i}'] = captured[(i, '_true')].astype(np.uint8)
        with torch.no_grad():
            p = model(x).double()[0]
        top2 = torch.topk(p, 2, dim=0).values
        out[f'margin_{i}'] = (top2[0] - top2[1]).numpy().astype(np.float32)
    save('g14_testing.npz', **out)


# ------------------------- G17: the reference's published inference size (README.md:10: 240 x 240 x 155 BraTS images), HNOSeg-XS forward
def g17_inference_full_size():
    """The reference's HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14)) (config_hnoseg_xs.ini) in eval mode under no_grad on one
    (1, 4, 240, 240, 155) formula volume -- the working grid is 121 x 121 x 78: 121 planes of 121 x 78, an even last axis, four partial
    row-pair tiles per plane: the shapes the round-5 item plane kernels were written for.  8 192 sampled probabilities, their sum per
    class, the arg-max labels at the samples with the top-2 margin (experiments/train_test.py:383-426 takes the arg max on the host),
    the label histogram; fp32 and with the reference run in float64."""
    torch.manual_seed(0)
    model = nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14)).eval()
    out = {}
    for k, v in model.state_dict().items():
        out[f'sd::{k}'] = v.detach().numpy().copy()
    shape = (1, 4, 240, 240, 155)
    x = T(formula_tensor(shape, 9))
    idx = None
    for tag in ('f32', 'f64'):
        m = model if tag == 'f32' else None
        if m is None:
            m = nets.HNOSegXS(4, 4, 24, [3] * 8, (10, 14, 14)).double().eval()
            m.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
        with torch.no_grad():
            y = m(x if tag == 'f32' else x.double())
        assert tuple(y.shape) == (1, 4) + shape[2:]
        if idx is None:
            idx = sample_indices(int(np.prod(shape[2:])), 8192, 3)       # voxel indices (all four classes of a voxel are kept)
            out['shape'], out['vox_idx'] = np.array(shape), idx
        probs = y.detach().double().reshape(4, -1)
        out[f'{tag}::probs'] = probs[:, idx].float().numpy()
        out[f'{tag}::class_sums'] = probs.sum(1).numpy()
        lab = probs.argmax(0)
        out[f'{tag}::labels'] = lab[idx].numpy().astype(np.uint8)
        out[f'{tag}::hist'] = torch.bincount(lab, minlength=4).numpy()
        top2 = probs[:, idx].topk(2, dim=0).values
        out[f'{tag}::margin'] = (top2[0] - top2[1]).float().numpy()
        del y, probs
    save('g17_inference_full_size.npz', **out)


if __name__ == '__main__':
    ALL = [g1_dht, g2_crop_pad, g3_operators, g4_mha, g5_losses, g6_hnosegxs, g6_128, g6s_small_models, g6b_xsblock_branch, g7_noseg_models,
           g7v_vnet_models, g7b_bf16_models, g15_cfg3_full_size, g16_cfg4_full_size, g9_misc, g10_two_d, g11_input, g12_mha_bias, g13_models_2d, g8_training, g14_testing, g17_inference_full_size]
    only = set(sys.argv[2:])   # e.g. `make_golden.py /root/reference g10_two_d` regenerates one fixture
    for fn in ALL:
        if not only or fn.__name__ in only:
            fn()
