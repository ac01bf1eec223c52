"""Pins the CPU oracle (oracle/hno_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from _inputs import formula_tensor, formula_labels, formula_volume, CROP_CASES, SMALL_MODELS
from oracle import hno_oracle as O

TOL32 = 2e-5   # fp32 FFT vs fp32 FFT on identical op order: rounding only
TOL64 = 1e-11
# dense fp64 formulation vs the reference's fp32 FFT: the FFT's rounding error scales with
# the FULL spectrum's energy while the kept block can be small, hence a looser bound
TOLD = 1e-4


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_dht_against_reference():
    g = load_golden('g1_dht.npz')
    for tag, shape in enumerate([(2, 3, 13, 15, 11), (1, 2, 33, 33, 33)]):
        for dt, tol in ((np.float32, TOL32), (np.float64, TOL64)):
            x = T(formula_tensor(shape, tag, dt))
            key = f's{tag}_{np.dtype(dt).name}'
            f3, i3, f2 = O.dht3(x).numpy(), O.dht3(x, True).numpy(), O.dht2(x).numpy()
            if tag == 0:
                assert rel_err(f3, g[f'{key}_fwd3']) < tol
                assert rel_err(i3, g[f'{key}_inv3']) < tol
                assert rel_err(f2, g[f'{key}_fwd2']) < tol
                assert rel_err(O.dht2(x, True).numpy(), g[f'{key}_inv2']) < tol
            else:
                idx = g[f'{key}_idx']
                assert rel_err(f3.ravel()[idx], g[f'{key}_fwd3']) < tol
                assert rel_err(i3.ravel()[idx], g[f'{key}_inv3']) < tol
                assert rel_err(f2.ravel()[idx], g[f'{key}_fwd2']) < tol


def test_dht_roundtrip_and_symmetry():
    x = T(formula_tensor((1, 2, 9, 7, 10), 3, np.float64))
    assert rel_err(O.dht3(O.dht3(x), True).numpy(), x.numpy()) < 1e-12
    # dense cas-matrix form == FFT form
    n = 10
    c = O.cas_matrix(n)
    assert torch.allclose(c, c.T)
    v = T(formula_tensor((n,), 1, np.float64))
    assert rel_err((c @ v / n).numpy(), O.dhtn(v, (-1,)).numpy()) < 1e-12


@pytest.mark.parametrize('ci', range(len(CROP_CASES)))
def test_transform_crop_pad_inverse(ci):
    g = load_golden('g2_crop_pad.npz')
    b, c, sp, modes = CROP_CASES[ci]
    k = f'c{ci}'
    x = T(formula_tensor((b, c) + sp, 10 + ci)).requires_grad_(True)
    z = O.transform_crop(x, modes)
    assert tuple(z.shape) == tuple(g[f'{k}_zshape'])
    assert rel_err(z.detach().numpy(), g[f'{k}_crop']) < TOL32
    cot = T(formula_tensor(tuple(z.shape), 20 + ci))
    (gx,) = torch.autograd.grad((z * cot).sum(), [x])
    assert rel_err(gx.numpy().ravel()[g[f'{k}_crop_gradx_idx']], g[f'{k}_crop_gradx']) < TOL32
    zin = T(formula_tensor(tuple(z.shape), 30 + ci)).requires_grad_(True)
    y = O.pad_inverse(zin, sp)
    assert rel_err(y.detach().numpy().ravel()[g[f'{k}_pad_idx']], g[f'{k}_pad']) < TOL32
    cot = T(formula_tensor(tuple(y.shape), 40 + ci))
    (gz,) = torch.autograd.grad((y * cot).sum(), [zin])
    assert rel_err(gz.numpy(), g[f'{k}_pad_gradz']) < TOL32
    # the pruned separable (dense) formulation the HIP kernels implement agrees too
    assert rel_err(O.dht_crop_dense(x.detach(), modes).numpy(), g[f'{k}_crop']) < TOLD
    assert rel_err(O.pad_idht_dense(zin.detach(), sp).numpy().ravel()[g[f'{k}_pad_idx']], g[f'{k}_pad']) < TOLD
    # adjoint identities used by the HIP backward
    n3 = float(np.prod(sp))
    assert rel_err((O.pad_idht_dense(cot_like(z, 20 + ci), sp) / n3).numpy().ravel()[g[f'{k}_crop_gradx_idx']],
                   g[f'{k}_crop_gradx']) < TOLD
    mm = O.clamp_modes(modes, sp)
    assert rel_err(O.dht_crop_dense(cot_like(y, 40 + ci), mm, scale=1.0).numpy(), g[f'{k}_pad_gradz']) < TOLD


def cot_like(t, tag):
    return T(formula_tensor(tuple(t.shape), tag))


def _operator_cases():
    case = 0
    for name in ('hartley', 'fourier'):
        for wt in ('shared', 'individual'):
            for use_transform in (True, False):
                for use_bias in (False, True):
                    yield name, wt, use_transform, use_bias, case
                    case += 1


@pytest.mark.parametrize('name,wt,use_transform,use_bias,case', list(_operator_cases()))
def test_operators(name, wt, use_transform, use_bias, case):
    g = load_golden('g3_operators.npz')
    ci_, sp, modes = 3, (12, 10, 14), (3, 2, 4)
    key = f'{name}_{wt}_t{int(use_transform)}_b{int(use_bias)}'
    assert int(g[f'{key}_case']) == case
    P = {k[len(key) + 3:]: T(g[k]).requires_grad_(True) for k in g.files if k.startswith(f'{key}_p_')}
    bias = P.get('bias')
    if use_transform:
        x = T(formula_tensor((2, ci_) + sp, 50)).requires_grad_(True)
    elif name == 'hartley':
        x = T(formula_tensor((2, ci_) + tuple(2 * m for m in modes), 70 + case)).requires_grad_(True)
    else:
        shp = (2, ci_, 2 * modes[0], 2 * modes[1], modes[2])
        x = torch.complex(T(formula_tensor(shp, 70 + case)), T(formula_tensor(shp, 170 + case))).requires_grad_(True)
    if name == 'hartley':
        y = O.hartley_operator(x, P['weight'], modes, bias, wt, use_transform)
        plist = [('weight', P['weight'])]
    else:
        if use_transform:
            y = O.fourier_operator(x, P['weight_real'], P['weight_imag'], modes, bias, wt)
        else:
            w = torch.complex(P['weight_real'], P['weight_imag'])
            eq = 'oi,bidhw->bodhw' if wt == 'shared' else 'oidhw,bidhw->bodhw'
            y = torch.einsum(eq, w, x)
            if bias is not None:
                y = y + bias
        plist = [('weight_real', P['weight_real']), ('weight_imag', P['weight_imag'])]
    if bias is not None:
        plist.append(('bias', bias))
    assert rel_err(_np(y), g[f'{key}_y']) < TOL32
    if y.is_complex():
        cot = torch.complex(T(formula_tensor(tuple(y.shape), 80 + case)), T(formula_tensor(tuple(y.shape), 180 + case)))
        loss = (y * cot.conj()).real.sum()
    else:
        loss = (y * T(formula_tensor(tuple(y.shape), 80 + case))).sum()
    gs = torch.autograd.grad(loss, [x] + [p for _, p in plist])
    assert rel_err(_np(gs[0]), g[f'{key}_gx']) < 5e-5
    for (pn, _), gp in zip(plist, gs[1:]):
        assert rel_err(_np(gp), g[f'{key}_g_{pn}']) < 5e-5, pn


def _np(t):
    return t.detach().numpy()


def test_losses():
    g = load_golden('g5_losses.npz')
    shape = (2, 4, 9, 10, 11)
    yp = torch.softmax(T(formula_tensor(shape, 90)), dim=1).requires_grad_(True)
    onehot = O.to_categorical(T(g['labels']), 4)
    assert onehot[1, 2].sum() == 0  # the eps path is really exercised
    assert rel_err(_np(O.corrcoef(yp, onehot)), g['corrcoef']) < 1e-5
    assert rel_err(_np(O.dice_coef(yp, onehot)), g['dice_coef']) < 1e-5
    for name, fn in (('pcc', O.pcc_loss), ('dice', O.dice_loss), ('expdice', O.exp_dice_loss)):
        val = fn(yp, onehot)
        (gr,) = torch.autograd.grad(val, [yp])
        assert abs(float(val) - float(g[f'{name}_loss'])) < 1e-6
        assert rel_err(gr.numpy(), g[f'{name}_grad']) < 1e-5


@pytest.mark.parametrize('tag', ['64', 'odd'])
def test_hnosegxs_full_model(tag):
    g = load_golden('g6_hnosegxs.npz')
    sd = {k[4:]: T(g[k]) for k in g.files if k.startswith('sd::')}
    assert sum(v.numel() for v in sd.values()) == 28248 == int(g['n_params'])
    shape = tuple(int(s) for s in g[f'{tag}_shape'])
    x = T(formula_tensor(shape, 7))
    lab = T(formula_labels((shape[0], 1) + shape[2:], 4, 5))
    y, loss, grads = O.hnosegxs_step(sd, x, lab, [3] * 8, (10, 14, 14))
    assert rel_err(y.numpy().ravel()[g[f'{tag}_y_idx']], g[f'{tag}_y']) < 1e-5
    assert abs(float(loss) - float(g[f'{tag}_loss'])) < 1e-6
    for k, gr in grads.items():
        assert rel_err(gr.numpy(), g[f'{tag}_grad::{k}']) < 1e-4, k


def test_labels_and_padcrop():
    g = load_golden('g9_misc.npz')
    lab = T(g['labels'])
    assert np.array_equal(O.to_categorical(lab, 5).numpy(), g['onehot5'])
    assert np.array_equal(O.to_categorical(lab).numpy(), g['onehot_auto'])
    mapping = {int(k): int(v) for k, v in zip(g['remap_keys'], g['remap_vals'])}
    assert np.array_equal(O.remap_labels(lab, mapping).numpy(), g['remapped'])
    x = T(formula_tensor((1, 2, 7, 8, 9), 3))
    for i, t in enumerate(g['padcrop_targets']):
        assert np.array_equal(O.spatial_padcrop(x, tuple(int(v) for v in t)).numpy(), g[f'padcrop_{i}'])


@pytest.mark.parametrize('name', list(SMALL_MODELS))
def test_small_models_oracle(name):
    g = load_golden('g6s_small_models.npz')
    kw, shape = SMALL_MODELS[name]
    pre = f'{name}::sd::'
    sd = {k[len(pre):]: T(g[k]) for k in g.files if k.startswith(pre)}
    K = kw['out_channels']
    x = T(formula_volume(shape, 3))
    lab = T(formula_labels((shape[0], 1) + shape[2:], K, 2))
    for lname in ('pcc', 'dice'):
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        y = O.hnosegxs_forward(params, x, kw['num_transform_blocks'], kw['num_modes'],
                               **{k: kw[k] for k in ('use_unet_skip', 'use_block_concat', 'use_deep_supervision', 'weights_type')
                                  if k in kw})
        loss = (O.pcc_loss if lname == 'pcc' else O.dice_loss)(y, O.to_categorical(lab, K))
        loss.backward()
        assert rel_err(_np(y), g[f'{name}::y']) < 1e-5
        assert abs(float(loss.detach()) - float(g[f'{name}::{lname}::loss'])) < 1e-6
        for k, p in params.items():
            assert rel_err(_np(p.grad), g[f'{name}::{lname}::grad::{k}']) < 1e-4, k


def test_xsblock_conv_branch_oracle():
    """HNOXSBlock(use_conv_branch=True) (golden G6b; nets/hnosegxs.py:185-329)."""
    from _inputs import XSBLOCK_BRANCH as cfg
    g = load_golden('g6b_xsblock_branch.npz')
    params = {'blk.' + k[4:]: T(g[k]).requires_grad_(True) for k in g.files if k.startswith('sd::')}
    x = T(formula_volume(cfg['shape'], 9)).requires_grad_(True)
    y = O.hnoxs_block(params, 'blk', x, cfg['num_modes'], cfg['num_convs'])
    cot = T(formula_tensor(tuple(y.shape), 61))
    (y * cot).sum().backward()
    assert rel_err(_np(y), g['y']) < 1e-5
    assert rel_err(_np(x.grad), g['gx']) < 1e-4
    for k, p in params.items():
        assert rel_err(_np(p.grad), g['grad::' + k[4:]]) < 1e-4, k


def test_g6_128_fixture_is_selfconsistent():
    """G6-128 (the metric's own grid): stored fp32 / float64 reference results agree to the fp32 noise floor."""
    g = load_golden('g6_128.npz')
    assert tuple(g['shape']) == (1, 4, 128, 128, 128) and g['y'].shape == g['y64'].shape
    assert rel_err(g['y'], g['y64']) < 2e-4 and abs(float(g['loss']) - float(g['loss64'])) < 1e-6
    n = sum(g[k].size for k in g.files if k.startswith('grad::'))
    assert n == 28248


from _inputs import NOSEG_MODELS  # noqa: E402


@pytest.mark.parametrize('name', list(NOSEG_MODELS))
def test_noseg_models_oracle(name):
    g = load_golden('g7_noseg_models.npz')
    kw, shape = NOSEG_MODELS[name]
    pre = f'{name}::sd::'
    params = {k[len(pre):]: T(g[k]).requires_grad_(True) for k in g.files if k.startswith(pre)}
    K = kw['out_channels']
    x = T(formula_volume(shape, 4))
    lab = T(formula_labels((shape[0], 1) + shape[2:], K, 6))
    y = O.neural_operator_seg_forward(params, x, kw['num_transform_blocks'], kw['num_modes'], kw['transform_type'],
                                      weights_type=kw.get('weights_type', 'shared'),
                                      use_block_skip=kw.get('use_block_skip', True),
                                      use_deep_supervision=kw.get('use_deep_supervision', False))
    loss = O.pcc_loss(y, O.to_categorical(lab, K))
    loss.backward()
    assert rel_err(_np(y), g[f'{name}::y']) < 1e-5
    assert abs(float(loss.detach()) - float(g[f'{name}::loss'])) < 1e-6
    for k, p in params.items():
        assert rel_err(_np(p.grad), g[f'{name}::grad::{k}']) < 1e-4, k


def test_cfg3_full_size_oracle():
    """BASELINE cfg3 at its real volume size (golden G15: the reference's FNOSeg, 24 Fourier blocks, on one 4 x 128^3 volume): the oracle's
    fp32 outputs, loss and all 71 184 gradients against the reference's fp32 run.  (~30 s of CPU time: the largest case of the CPU suite.)"""
    g = load_golden('g15_cfg3_full_size.npz')
    params = {k[4:]: T(g[k]).requires_grad_(True) for k in g.files if k.startswith('sd::')}
    shape = tuple(int(v) for v in g['shape'])
    x = T(formula_tensor(shape, 7))
    lab = T(formula_labels((1, 1) + shape[2:], 4, 5))
    y = O.neural_operator_seg_forward(params, x, 24, (10, 14, 14), 'Fourier')
    loss = O.pcc_loss(y, O.to_categorical(lab, 4))
    loss.backward()
    assert rel_err(_np(y).ravel()[g['y_idx']], g['f32::y']) < 2e-5
    assert abs(float(y.detach().double().sum()) - float(g['f32::y_sum'])) / float(g['f32::y_sum']) < 1e-6
    assert abs(float(loss.detach()) - float(g['f32::loss'])) < 1e-6
    num = den = 0.0
    for k, p in params.items():
        if f'f32::grad::{k}' in g.files:
            ref = g[f'f32::grad::{k}'].astype(np.float64)
            num += ((_np(p.grad).astype(np.float64) - ref) ** 2).sum()
            den += (ref ** 2).sum()
    assert den > 0 and np.sqrt(num / den) < 1e-4


def test_inference_full_size_oracle():
    """The reference's published inference size (README.md:10; golden G17: its HNOSeg-XS in eval mode on one 4 x 240 x 240 x 155 volume,
    working grid 121 x 121 x 78): the oracle's probabilities and arg-max labels against the reference's fp32 run.  (~25 s of CPU time.)"""
    g = load_golden('g17_inference_full_size.npz')
    params = {k[4:]: T(g[k]) for k in g.files if k.startswith('sd::')}
    shape = tuple(int(v) for v in g['shape'])
    x = T(formula_tensor(shape, 9))
    with torch.no_grad():
        y = O.hnosegxs_forward(params, x, [3] * 8, (10, 14, 14))
    probs = y.double().reshape(4, -1)
    idx = g['vox_idx']
    assert rel_err(probs[:, idx].float().numpy(), g['f32::probs']) < 2e-5
    assert np.abs(probs.sum(1).numpy() - g['f32::class_sums']).max() / g['f32::class_sums'].max() < 1e-6
    lab = probs.argmax(0)[idx].numpy()
    sure = g['f32::margin'] > 1e-4
    assert sure.mean() > 0.9 and np.array_equal(lab[sure], g['f32::labels'][sure])


def test_cfg4_full_size_oracle():
    """BASELINE cfg4 at its real volume size (golden G16: the reference's V-Net-DS on one 4 x 160 x 192 x 128 volume): the oracle's fp32
    outputs, loss and sampled gradients against the reference's fp32 run, on the weights the constructor draws under seed 0 (proved equal to
    the reference's by the fixture's per-tensor sums).  (~40 s of CPU time.)"""
    import multimodal_3d_image_segmentation_amd as pkg
    from _inputs import sample_indices
    g = load_golden('g16_cfg4_full_size.npz')
    torch.manual_seed(0)
    model = pkg.nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4])
    for i, (k, p) in enumerate(model.named_parameters()):
        assert abs(float(p.detach().double().sum()) - float(g['param_sum'][i])) <= 1e-9 * max(1.0, abs(float(g['param_sum'][i]))), k
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    shape = tuple(int(v) for v in g['shape'])
    x = T(formula_tensor(shape, 9))
    lab = T(formula_labels((1, 1) + shape[2:], 4, 3))
    y = O.vnetds_forward(params, x, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4])
    loss = O.pcc_loss(y, O.to_categorical(lab, 4))
    loss.backward()
    assert rel_err(_np(y).ravel()[g['y_idx']], g['f32::y']) < 2e-5
    assert abs(float(loss.detach()) - float(g['f32::loss'])) < 1e-6
    samples = np.concatenate([_np(params[k].grad).ravel()[sample_indices(params[k].numel(), 512, 5)] for k in [str(v) for v in g['param_names']]])
    ref = g['f32::grad_samples'].astype(np.float64)
    assert np.sqrt(((samples.astype(np.float64) - ref) ** 2).sum() / (ref ** 2).sum()) < 1e-4


from _inputs import MHA_CASES  # noqa: E402


@pytest.mark.parametrize('ci', range(len(MHA_CASES)))
def test_hartley_mha_oracle(ci):
    g = load_golden('g4_mha.npz')
    cin, kd, heads, modes, patch, nin = MHA_CASES[ci]
    k = f'm{ci}'
    shape = (1, 6, 12, 14, 12)
    xs = [T(formula_tensor(shape, 90 + ci + 7 * j)).requires_grad_(True) for j in range(nin)]
    P = {n: T(g[f'{k}_p_{n}']).requires_grad_(True) for n in ('weight_query', 'weight_key', 'weight_value', 'weight_out')}
    y = O.hartley_mha(xs[0], P['weight_query'], P['weight_key'], P['weight_value'], P['weight_out'], modes, patch,
                      x_key=xs[1] if nin >= 2 else None, x_value=xs[2] if nin == 3 else None)
    assert rel_err(_np(y), g[f'{k}_y']) < 2e-5
    cot = T(formula_tensor(tuple(y.shape), 95 + ci))
    gs = torch.autograd.grad((y * cot).sum(), xs + list(P.values()))
    for j in range(nin):
        assert rel_err(_np(gs[j]), g[f'{k}_gx{j}']) < 1e-4
    for (pn, _), gp in zip(P.items(), gs[nin:]):
        assert rel_err(_np(gp), g[f'{k}_g_{pn}']) < 1e-4, pn


from _inputs import VNET_MODELS  # noqa: E402


@pytest.mark.parametrize('name', list(VNET_MODELS))
def test_vnet_models_oracle(name):
    g = load_golden('g7v_vnet_models.npz')
    kw, shape = VNET_MODELS[name]
    pre = f'{name}::sd::'
    params = {k[len(pre):]: T(g[k]).requires_grad_(True) for k in g.files if k.startswith(pre)}
    K = kw['out_channels']
    x = T(formula_volume(shape, 5))
    lab = T(formula_labels((shape[0], 1) + shape[2:], K, 7))
    y = O.vnetds_forward(params, x, kw['num_blocks'], kw.get('right_leg_indexes'), kw.get('use_resize', True))
    loss = O.dice_loss(y, O.to_categorical(lab, K))
    loss.backward()
    assert rel_err(_np(y), g[f'{name}::y']) < 1e-5
    assert abs(float(loss.detach()) - float(g[f'{name}::loss'])) < 1e-6
    for k, p in params.items():
        assert rel_err(_np(p.grad), g[f'{name}::grad::{k}']) < 1e-4, k


def test_input_normalisation_oracle_vs_reference():
    """oracle.normalize_modalities against the reference's numpy masked-array implementation (golden G11)."""
    from _inputs import raw_modalities
    g = load_golden('g11_input.npz')
    vol = raw_modalities()
    assert rel_err(O.normalize_modalities(vol, mask_val=0), g['norm_masked']) < 2e-6
    assert rel_err(O.normalize_modalities(vol), g['norm_plain']) < 2e-6
    assert rel_err(O.normalize_modalities(vol, mask_val=0, clip_val=(0, 600)), g['norm_clip']) < 2e-6
    assert np.all(O.normalize_modalities(vol, mask_val=0)[vol == 0] == 0)


def test_affine_nearest_oracle_properties():
    """The resampling oracle is not pinned by a reference run (no SimpleITK here); check what must hold for any
    nearest-neighbour resampler: identity, integer shifts, axis flips and the constant fill."""
    x = formula_tensor((2, 5, 6, 7), 9)
    eye = np.eye(4)
    assert np.array_equal(O.affine_nearest(x, O.centre_affine(eye, x.shape[1:])), x)
    sh = np.eye(4)
    sh[:3, 3] = (2, -1, 1)          # (x, y, z): out[d, h, w] = in[d + 1, h - 1, w + 2]
    got = O.affine_nearest(x, O.centre_affine(sh, x.shape[1:]), cval=-7.0)
    want = np.full_like(x, -7.0)
    want[:, :4, 1:, :5] = x[:, 1:, :5, 2:]
    assert np.array_equal(got, want)
    assert np.array_equal(O.affine_nearest(x, O.centre_affine(eye, x.shape[1:]), flips=(1, 3)), x[:, ::-1, :, ::-1])
    x2 = formula_tensor((1, 6, 8), 2)
    assert np.array_equal(O.affine_nearest(x2, O.centre_affine(np.eye(3), x2.shape[1:])), x2)


def test_affine_nearest_oracle_vs_scipy_resampler():
    """Round-4 verdict item 9: the resampling oracle against an INDEPENDENT implementation that ships in the image,
    ``scipy.ndimage.affine_transform(order=0)`` -- still "parity unpinned by the reference" (SimpleITK is absent, the reference's
    ResampleImageFilter call, dataset.py:202-237, cannot run here), but no longer checked only against itself.
    Conventions written out: the oracle's rows [M | t] map OUTPUT voxel (x, y, z) to the continuous INPUT index (x, y, z)' = M (x, y, z) + t
    (dataset.py:218-237: SimpleITK's AffineTransform about the image centre, unit spacing, zero origin, identity direction);
    scipy maps output ARRAY index (z, y, x) through ``matrix @ o + offset``, so matrix = M with rows and columns reversed and
    offset = t reversed.  Border rule: ITK's nearest-neighbour interpolator accepts a continuous index c iff -0.5 <= c < size - 0.5 and
    takes floor(c + 0.5); scipy's order-0 spline takes floor(c + 0.5) too (ni_interpolation.c) and, with mode='grid-constant', fills
    exactly when that index falls outside the array -- the same set.  (mode='constant' fills already for c < 0 or c > size - 1: it
    agrees on every voxel whose input index lies inside [0, size - 1], checked as well.)
    Matrices and offsets are dyadic rationals, so both implementations compute every coordinate exactly and ties (c + 0.5 integral)
    must agree too; a generic rotation + anisotropic scaling case allows the few voxels whose coordinate lands within 1e-9 of a tie."""
    from scipy import ndimage
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 9, 11, 13)).astype(np.float32)

    def scipy_resample(m12, mode, cval):
        m12 = np.asarray(m12, dtype=np.float64)
        A, off = m12[:, :3][::-1, ::-1], m12[:, 3][::-1]
        return np.stack([ndimage.affine_transform(x[c], A, offset=off, order=0, mode=mode, cval=cval) for c in range(x.shape[0])])

    def coords(m12):
        D, H, W = x.shape[1:]
        z, y, xx = np.meshgrid(np.arange(D, dtype=np.float64), np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
        m = np.asarray(m12, dtype=np.float64)
        return [m[i, 0] * xx + m[i, 1] * y + m[i, 2] * z + m[i, 3] for i in range(3)], (W, H, D)
    filled = 0
    for trial in range(24):
        M = np.round(rng.uniform(-1.5, 1.5, (3, 3)) * 8) / 8
        t = np.round(rng.uniform(-3, 8, 3) * 16) / 16
        m12 = np.concatenate([M, t[:, None]], 1)
        ref = O.affine_nearest(x, m12, cval=-7.0)
        assert np.array_equal(scipy_resample(m12, 'grid-constant', -7.0), ref), trial
        cs, sizes = coords(m12)
        interior = np.ones(x.shape[1:], dtype=bool)
        for c, n in zip(cs, sizes):
            interior &= (c >= 0) & (c <= n - 1)
        assert np.array_equal(scipy_resample(m12, 'constant', -7.0)[:, interior], ref[:, interior]), trial
        filled += int((ref == -7.0).sum())
    assert filled > 1000                                    # the fill rule was really exercised
    # the transforms ImageTransform draws (rotation, anisotropic zoom, shift about the centre: dataset.py:95-199)
    for trial in range(6):
        ang = rng.uniform(-0.5, 0.5, 3)
        cz, sz, cy, sy, cx, sx = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
        R = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        full = np.eye(4)
        full[:3, :3] = R @ np.diag(rng.uniform(0.8, 1.25, 3))
        full[:3, 3] = rng.uniform(-2, 2, 3)
        m12 = O.centre_affine(full, x.shape[1:])
        ref = O.affine_nearest(x, m12, cval=0.5)
        got = scipy_resample(m12, 'grid-constant', 0.5)
        assert (got != ref).mean() < 1e-3, trial
