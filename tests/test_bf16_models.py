"""bf16 (autocast) model-level parity: golden G7b = the REFERENCE run under torch.autocast(bfloat16) next to its own fp32 run.

Tolerance.  bf16 keeps 8 significant bits; through a network the reference's own bf16 results sit 0.6-1.4e-2 (outputs) and
1.5-18 % (gradients, mean over tensors of max-relative error) away from its fp32 results -- that distance, not 1e-4, is the
resolution at which a second bf16 implementation can be compared.  Stated bars:
  * outputs: within 2e-2 (relative to max) of the reference's fp32 outputs, and no further from them than 2x the reference's
    own bf16 run is;
  * loss: within 2e-3 absolute of the reference's fp32 loss;
  * gradients: relative L2 error of the WHOLE gradient against the reference's fp32 gradient <= max(5e-2, 2x the reference's
    own bf16-vs-fp32 L2 distance)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from _inputs import BF16_MODELS, formula_volume, formula_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pkg():
    import multimodal_3d_image_segmentation_amd as p
    p._lib.lib()
    assert torch.cuda.is_available()
    return p


def _run(pkg, name, autocast=True):
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    g = load_golden('g7b_bf16_models.npz')
    cls, kw, shape = BF16_MODELS[name]
    model = getattr(pkg.nets, cls)(**kw)
    pre = f'{name}::sd::'
    model.load_state_dict({k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)})
    model = model.cuda()
    K = kw['out_channels']
    x = torch.from_numpy(formula_volume(shape, 6)).cuda()
    lab = torch.from_numpy(formula_labels((shape[0], 1) + shape[2:], K, 8)).cuda()
    u8 = pkg.ops.labels_prepare(lab, K)
    import contextlib
    with (torch.autocast('cuda', dtype=torch.bfloat16) if autocast else contextlib.nullcontext()):
        y = model(x)
        loss = custom_losses.PCCLoss()(y, u8)
    loss.backward()
    return g, model, y, loss


def _l2(model, g, name, tag_ref='f32'):
    num = den = 0.0
    for k, p in model.named_parameters():
        ref = g[f'{name}::{tag_ref}::grad::{k}'].astype(np.float64)
        num += ((p.grad.cpu().numpy().astype(np.float64) - ref) ** 2).sum()
        den += (ref ** 2).sum()
    return float(np.sqrt(num / den))


def _l2_ref(g, name):
    num = den = 0.0
    for k in g.files:
        if k.startswith(f'{name}::bf16::grad::'):
            ref = g[k.replace('::bf16::', '::f32::')].astype(np.float64)
            num += ((g[k].astype(np.float64) - ref) ** 2).sum()
            den += (ref ** 2).sum()
    return float(np.sqrt(num / den))


@pytest.mark.parametrize('name', [n for n in BF16_MODELS if n.startswith('vnet')])
def test_vnet_bf16_vs_reference_autocast_golden(pkg, name):
    from multimodal_3d_image_segmentation_amd import ops_bf16
    before = dict(ops_bf16._stats)
    g, model, y, loss = _run(pkg, name)
    # every convolution that feeds a GroupNorm takes its bias gradient from the GroupNorm backward's reductions (no pass over dy);
    # the conv bias gradients are then judged with all the others against the reference's
    assert ops_bf16._stats['colsum_fused'] > before['colsum_fused']
    bias_keys = [k for k, p in model.named_parameters() if k.endswith('op.bias')]
    for k in bias_keys[:8]:
        ref = g[f'{name}::f32::grad::{k}']
        got = dict(model.named_parameters())[k].grad.cpu().numpy()
        assert np.abs(got - ref).max() <= 5e-2 * max(np.abs(ref).max(), 1e-6) + 1e-6, k
    yv = y.detach().float().cpu().numpy()
    d_ref = rel_err(g[f'{name}::bf16::y'], g[f'{name}::f32::y'])
    d = rel_err(yv, g[f'{name}::f32::y'])
    l2, l2_ref = _l2(model, g, name), _l2_ref(g, name)
    print(f'{name}: outputs vs reference fp32 {d:.2e} (reference bf16: {d_ref:.2e}); loss {float(loss):.6f} vs '
          f'{float(g[f"{name}::f32::loss"]):.6f}; gradient L2 vs reference fp32 {l2:.2e} (reference bf16: {l2_ref:.2e})')
    assert np.isfinite(yv).all()
    assert d < 2e-2 and d < max(5e-3, 2.0 * d_ref)
    assert abs(float(loss) - float(g[f'{name}::f32::loss'])) < 2e-3
    assert l2 < max(5e-2, 2.0 * l2_ref)
    for k, p in model.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
    # the model's per-step packed-weight buffers are only visible during its own forward (a later tensor may reuse a freed
    # parameter's address: a stale look-up once fed another layer's weights to an unrelated convolution)
    from multimodal_3d_image_segmentation_amd import ops_bf16
    assert ops_bf16.PackedWeights._current is None
    assert ops_bf16.PackedWeights.lookup(next(model.parameters())) is None


@pytest.mark.parametrize('name', [n for n in BF16_MODELS if not n.startswith('vnet')])
def test_spectral_models_bf16_vs_reference_autocast_golden(pkg, name):
    """FNOSeg / HNOSeg / HNOSeg-XS under autocast (BASELINE cfg3's family): the reference keeps transforms and the complex mix in fp32
    and runs the spatial 1x1x1 convolutions in bf16; here those convolutions take the bf16 matrix-core variants of the pointwise
    kernels at the BASELINE channel count (24; other widths keep fp32 arithmetic, i.e. are closer to the fp32 run).  Same bars
    as the V-Net test: outputs no further from the reference's fp32 outputs than 2x its own bf16 run, whole-gradient L2 error
    <= max(5e-2, 2x the reference's own bf16-vs-fp32 distance)."""
    g, model, y, loss = _run(pkg, name)
    yv = y.detach().float().cpu().numpy()
    d_ref = rel_err(g[f'{name}::bf16::y'], g[f'{name}::f32::y'])
    d = rel_err(yv, g[f'{name}::f32::y'])
    l2, l2_ref = _l2(model, g, name), _l2_ref(g, name)
    print(f'{name}: outputs vs reference fp32 {d:.2e} (reference bf16: {d_ref:.2e}); loss {float(loss.detach()):.6f} vs '
          f'{float(g[f"{name}::f32::loss"]):.6f}; gradient L2 vs reference fp32 {l2:.2e} (reference bf16: {l2_ref:.2e})')
    assert np.isfinite(yv).all()
    assert d < 2.5e-2 and d < max(5e-3, 2.0 * d_ref)
    assert abs(float(loss.detach()) - float(g[f'{name}::f32::loss'])) < 2e-3
    assert l2 < max(5e-2, 2.0 * l2_ref)
    if '24' in name:      # the bf16 kernels really ran: the result differs from the fp32 kernels' by bf16-sized amounts
        g2, model2, y2, loss2 = _run(pkg, name, autocast=False)
        assert rel_err(y2.detach().cpu().numpy(), g[f'{name}::f32::y']) < 1e-4
        assert rel_err(yv, y2.detach().cpu().numpy()) > 1e-4


def test_fnoseg_cfg3_full_size_bf16_step(pkg):
    """BASELINE cfg3 at its real size (FNOSeg, 24 blocks, 2 x 4 x 128^3) under autocast: finite, loss within 2e-3 of the fp32
    kernels' loss, gradient cosine with the fp32 gradient > 0.96."""
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    import contextlib
    torch.manual_seed(0)
    model = pkg.nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier').cuda()
    gen = torch.Generator(device='cuda').manual_seed(6)
    x = torch.randn((2, 4, 128, 128, 128), device='cuda', generator=gen)
    lab = torch.randint(0, 4, (2, 1, 128, 128, 128), device='cuda', generator=gen).float()
    u8 = pkg.ops.labels_prepare(lab, 4)
    res = {}
    for tag in ('bf16', 'f32'):
        for p in model.parameters():
            p.grad = None
        with (torch.autocast('cuda', dtype=torch.bfloat16) if tag == 'bf16' else contextlib.nullcontext()):
            y = model(x)
            loss = custom_losses.PCCLoss()(y, u8)
        loss.backward()
        assert torch.isfinite(y).all()
        res[tag] = (float(loss.detach()), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double())
        del y, loss
    (lb, gb), (lf, gf) = res['bf16'], res['f32']
    cos = float((gb * gf).sum() / (gb.norm() * gf.norm()))
    print(f'cfg3 full size: loss bf16 {lb:.6f} fp32 {lf:.6f}; gradient cosine {cos:.5f}, norm ratio {float(gb.norm() / gf.norm()):.4f}')
    assert torch.isfinite(gb).all() and abs(lb - lf) < 2e-3
    # 24 blocks of bf16 convolutions between fp32 transforms: measured cosine 0.981, norm ratio 0.981
    assert cos > 0.96 and 0.9 < float(gb.norm() / gf.norm()) < 1.1 and float((gb - gf).abs().max()) > 0.0


@pytest.mark.parametrize('batch', [1, 2])
def test_fnoseg_cfg3_full_size_vs_reference_golden(pkg, batch):
    """BASELINE cfg3 at its REAL size against the REFERENCE (golden G15: the reference's FNOSeg, 24 Fourier blocks, on one 4 x 128^3 volume
    in fp32, under torch.autocast('cpu', bfloat16) and in float64; batch 2 = the same volume stacked, the bench's shape).  Round 4 compared
    the bf16 kernels with the repo's own fp32 kernels at this size (the test above).  The float64 run is the truth: at this depth the
    reference's own fp32 gradients are 1.4e-3 (L2) from it and its own bf16 run 5.5e-2 in the outputs / 0.36 in the gradients.
      fp32 kernels: sampled outputs < 1e-4 and loss < 1e-5 of the reference's fp32 run; gradient error against float64 no larger than
                    2x the reference's own fp32 error (the bar of the 128^3 HNOSeg-XS test);
      bf16 kernels: outputs and gradients no further from float64 than 1.5x the reference's own bf16 run, loss within 2e-3."""
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    from _inputs import formula_tensor
    import contextlib
    g = load_golden('g15_cfg3_full_size.npz')
    model = pkg.nets.NeuralOperatorSeg(4, 4, 24, 24, (10, 14, 14), 'Fourier')
    model.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd::')})
    model = model.cuda()
    shape = tuple(int(v) for v in g['shape'])
    x = torch.from_numpy(formula_tensor(shape, 7)).cuda().expand(batch, *shape[1:]).contiguous()
    lab = torch.from_numpy(formula_labels((1, 1) + shape[2:], 4, 5)).cuda().expand(batch, 1, *shape[2:]).contiguous()
    u8 = pkg.ops.labels_prepare(lab, 4)
    keys = [k for k, _ in model.named_parameters()]

    def l2(get, tag_ref='f64'):
        num = den = 0.0
        for k in keys:
            ref = g[f'{tag_ref}::grad::{k}'].astype(np.float64)
            num += ((get(k).astype(np.float64) - ref) ** 2).sum()
            den += (ref ** 2).sum()
        return float(np.sqrt(num / den))
    ref_err = {t: (rel_err(g[f'{t}::y'], g['f64::y']), l2(lambda k, t=t: g[f'{t}::grad::{k}'])) for t in ('f32', 'bf16')}
    for tag in ('f32', 'bf16'):
        for p in model.parameters():
            p.grad = None
        with (torch.autocast('cuda', dtype=torch.bfloat16) if tag == 'bf16' else contextlib.nullcontext()):
            y = model(x)
            loss = custom_losses.PCCLoss()(y, u8)
        loss.backward()
        grads = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
        err = l2(lambda k: grads[k])
        for b in range(batch):
            yv = y[b].detach().float().cpu().numpy().ravel()[g['y_idx']]
            d32, d64 = rel_err(yv, g['f32::y']), rel_err(yv, g['f64::y'])
            if tag == 'f32':
                assert d32 < 1e-4 and d64 < 1e-4, (d32, d64)
                assert abs(float(y[b].double().sum()) - float(g['f32::y_sum'])) / float(g['f32::y_sum']) < 1e-6
            else:
                assert d64 < 1.5 * ref_err['bf16'][0], (d64, ref_err['bf16'][0])
        print(f'cfg3 full size B={batch} {tag}: outputs vs float64 {d64:.2e} (reference {tag}: {ref_err[tag][0]:.2e}); loss {float(loss.detach()):.6f} vs '
              f'{float(g["f64::loss"]):.6f}; gradient L2 vs float64 {err:.2e} (reference {tag}: {ref_err[tag][1]:.2e})')
        if tag == 'f32':
            assert abs(float(loss.detach()) - float(g['f32::loss'])) < 1e-5
            assert err < max(1e-4, 2.0 * ref_err['f32'][1])
        else:
            assert abs(float(loss.detach()) - float(g['f64::loss'])) < 2e-3
            assert err < 1.5 * ref_err['bf16'][1]
        del y, loss


def test_vnet_fp32_path_unchanged_outside_autocast(pkg):
    """the same model without autocast still runs the fp32 kernels and matches the reference's fp32 run at 1e-4 / 2e-4"""
    name = 'vnet_ds_bf16'
    g, model, y, loss = _run(pkg, name, autocast=False)
    assert rel_err(y.detach().cpu().numpy(), g[f'{name}::f32::y']) < 1e-4
    assert abs(float(loss) - float(g[f'{name}::f32::loss'])) < 1e-5
    assert _l2(model, g, name) < 2e-4


def test_fp16_autocast_is_refused(pkg):
    cls, kw, shape = BF16_MODELS['vnet_ds_bf16']
    model = pkg.nets.VNetDS(**kw).cuda()
    with pytest.raises(NotImplementedError), torch.autocast('cuda', dtype=torch.float16):
        model(torch.zeros(shape, device='cuda'))


def test_vnet_cfg4_full_size_bf16_step(pkg):
    """BASELINE cfg4 at its real size (V-Net-DS 22.5 M parameters, 1 x 4 x 160 x 192 x 128): the bf16 step is finite, its loss
    agrees with the fp32 kernels' loss on the same weights and input to 2e-3, and the bf16 gradient points the same way as the
    fp32 one (cosine > 0.98 over all 22.5 M components)."""
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    torch.manual_seed(0)
    model = pkg.nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4]).cuda()
    gen = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn((1, 4, 160, 192, 128), device='cuda', generator=gen)
    lab = torch.randint(0, 4, (1, 1, 160, 192, 128), device='cuda', generator=gen).float()
    u8 = pkg.ops.labels_prepare(lab, 4)
    res = {}
    for tag in ('bf16', 'f32'):
        for p in model.parameters():
            p.grad = None
        import contextlib
        with (torch.autocast('cuda', dtype=torch.bfloat16) if tag == 'bf16' else contextlib.nullcontext()):
            y = model(x)
            loss = custom_losses.PCCLoss()(y, u8)
        loss.backward()
        assert torch.isfinite(y).all()
        s = y.sum(dim=1)
        assert float((s - 1).abs().max()) < 1e-4                       # softmax output
        res[tag] = (float(loss), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double())
        del y, loss, s
    (lb, gb), (lf, gf) = res['bf16'], res['f32']
    cos = float((gb * gf).sum() / (gb.norm() * gf.norm()))
    print(f'cfg4 full size: loss bf16 {lb:.6f} fp32 {lf:.6f}; gradient cosine {cos:.5f}, norm ratio {float(gb.norm() / gf.norm()):.4f}')
    assert torch.isfinite(gb).all() and abs(lb - lf) < 2e-3
    assert cos > 0.98 and 0.9 < float(gb.norm() / gf.norm()) < 1.1


def test_vnet_cfg4_full_size_vs_reference_golden(pkg):
    """BASELINE cfg4 at its REAL size against the REFERENCE (golden G16: the reference's V-Net-DS, 22.5 M parameters, on one
    4 x 160 x 192 x 128 volume in fp32, under torch.autocast('cpu', bfloat16) and in float64).  The weights are the constructor's under
    torch.manual_seed(0) -- this package draws the same values as the reference, which the fixture's per-tensor sums and leading elements
    prove first.  Gradients: 512 sampled elements and the L2 norm per parameter tensor (161 tensors).  Bars against the float64 run:
      fp32 kernels: outputs < 1e-4, loss < 1e-5, sampled-gradient L2 error < 1e-4 + 2x the reference's own fp32 error;
      bf16 kernels: outputs / sampled gradients / per-tensor norms no further from float64 than 1.5x the reference's own bf16 run."""
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    from _inputs import formula_tensor, sample_indices
    import contextlib
    g = load_golden('g16_cfg4_full_size.npz')
    torch.manual_seed(0)
    model = pkg.nets.VNetDS(4, 4, 24, [1, 2, 3, 3, 3], right_leg_indexes=[0, 1, 2, 3, 4])
    names = [k for k, _ in model.named_parameters()]
    assert names == [str(v) for v in g['param_names']]
    for i, (k, p) in enumerate(model.named_parameters()):          # the same initial weights as the reference drew
        assert np.array_equal(np.resize(p.detach().numpy().ravel()[:8], 8), g['param_head'][i]), k
        assert abs(float(p.detach().double().sum()) - float(g['param_sum'][i])) <= 1e-9 * max(1.0, abs(float(g['param_sum'][i]))), k
    model = model.cuda()
    shape = tuple(int(v) for v in g['shape'])
    x = torch.from_numpy(formula_tensor(shape, 9)).cuda()
    lab = torch.from_numpy(formula_labels((1, 1) + shape[2:], 4, 3)).cuda()
    u8 = pkg.ops.labels_prepare(lab, 4)
    gidx = [sample_indices(p.numel(), 512, 5) for p in model.parameters()]
    assert [len(i) for i in gidx] == [int(v) for v in g['grad_sample_counts']]
    truth = g['f64::grad_samples'].astype(np.float64)

    def errs(samples, norms):
        e = float(np.sqrt(((samples.astype(np.float64) - truth) ** 2).sum() / (truth ** 2).sum()))
        n = float(np.abs(norms - g['f64::grad_norm']).max() / g['f64::grad_norm'].max())
        return e, n
    ref = {t: (rel_err(g[f'{t}::y'], g['f64::y']),) + errs(g[f'{t}::grad_samples'], g[f'{t}::grad_norm']) for t in ('f32', 'bf16')}
    for tag in ('f32', 'bf16'):
        for p in model.parameters():
            p.grad = None
        with (torch.autocast('cuda', dtype=torch.bfloat16) if tag == 'bf16' else contextlib.nullcontext()):
            y = model(x)
            loss = custom_losses.PCCLoss()(y, u8)
        loss.backward()
        yv = y.detach().float().cpu().numpy().ravel()[g['y_idx']]
        d = rel_err(yv, g['f64::y'])
        samples = np.concatenate([p.grad.cpu().numpy().ravel()[i] for p, i in zip(model.parameters(), gidx)])
        norms = np.array([float(p.grad.double().norm()) for p in model.parameters()])
        e, n = errs(samples, norms)
        print(f'cfg4 full size {tag}: outputs vs float64 {d:.2e} (reference {tag}: {ref[tag][0]:.2e}); loss {float(loss.detach()):.6f} vs '
              f'{float(g["f64::loss"]):.6f}; sampled-gradient L2 vs float64 {e:.2e} (reference: {ref[tag][1]:.2e}); per-tensor norms {n:.2e} '
              f'(reference: {ref[tag][2]:.2e})')
        if tag == 'f32':
            assert d < 1e-4 and abs(float(loss.detach()) - float(g['f64::loss'])) < 1e-5
            assert abs(float(y.double().sum()) - float(g['f32::y_sum'])) / float(g['f32::y_sum']) < 1e-6
            assert e < 1e-4 + 2.0 * ref['f32'][1] and n < 1e-4 + 2.0 * ref['f32'][2]
        else:
            assert d < 1.5 * ref['bf16'][0] and abs(float(loss.detach()) - float(g['f64::loss'])) < 2e-3
            assert e < min(5e-2, 1.5 * ref['bf16'][1]) and n < min(1e-2, 1.5 * ref['bf16'][2])      # (measured 2.4e-2 / 1.7e-3; the reference's CPU bf16 run: 0.71 / 0.18)
        del y, loss


def test_vnet_2d_model_under_autocast_runs_the_fp32_kernels(pkg):
    """A 2-D V-Net-DS (ndim = 4) under bf16 autocast used to raise; since round 6 it runs the fp32 kernels (autocast only ever lowers
    precision): same output as without autocast."""
    torch.manual_seed(4)
    model = pkg.nets.VNetDS(2, 3, 8, [1, 1], right_leg_indexes=[0, 1], ndim=4).cuda()
    x = torch.randn(2, 2, 32, 32, device='cuda')
    y0 = model(x)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y1 = model(x)
    assert y1.dtype == torch.float32 and torch.equal(y0, y1)
