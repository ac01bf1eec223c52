"""GPU parity tests of the bf16 matrix-core path (hno_cb_*): every op against a float64 torch computation on the SAME
bf16-rounded operands, so the only differences are the fp32 accumulation order and the final rounding to bf16.

Tolerances (relative to max): bf16 has 8 significant bits (eps = 2^-8 = 3.9e-3); a bf16 OUTPUT is therefore held to 4e-3
(one rounding of a value near the maximum), fp32 outputs (weight / bias / affine gradients, statistics) to 2e-5 .. 1e-4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

BF16_TOL = 4e-3


@pytest.fixture(scope='module')
def ob():
    import multimodal_3d_image_segmentation_amd as p
    p._lib.lib()
    assert torch.cuda.is_available()
    from multimodal_3d_image_segmentation_amd import ops_bf16
    return ops_bf16


def rb(t):
    """round to bf16, keep fp32 container"""
    return t.bfloat16().float()


def to_cl(x):
    """(B, C, D, H, W) fp32 (already bf16-representable) -> channels-last bf16 on the GPU"""
    return x.permute(0, 2, 3, 4, 1).contiguous().bfloat16().cuda()


def from_cl(y):
    return y.detach().float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def test_pack_unpack_roundtrip(ob):
    torch.manual_seed(0)
    x = torch.randn(2, 4, 5, 6, 7)
    y = ob.pack_input_raw(x.cuda(), 8)
    assert y.shape == (2, 5, 6, 7, 8) and y.dtype == torch.bfloat16
    assert torch.equal(from_cl(y)[:, :4], rb(x)) and float(from_cl(y)[:, 4:].abs().max()) == 0.0
    back = ob.unpack_raw(y, 4)
    assert torch.equal(back.cpu(), rb(x))


CONV_CASES = [
    # (B, Ca, Cb, Cout, spatial, ks, stride, transposed)
    (1, 24, 0, 24, (9, 10, 11), 3, 1, False),
    (2, 24, 24, 24, (7, 9, 13), 3, 1, False),       # decoder conv: two concatenated inputs
    (1, 48, 0, 48, (6, 7, 9), 3, 1, False),
    (1, 48, 0, 96, (5, 6, 5), 3, 1, False),
    (1, 24, 0, 24, (9, 11, 13), 3, 2, False),       # strided down-convolution
    (1, 48, 0, 24, (5, 6, 7), 3, 2, True),          # ConvTranspose3d k3 s2 p1 op1
    # round 4: stride-2 fractional gathers run by parity class (only the 1 / 2 / 4 / 8 valid taps of a class are visited): transposed
    # convolutions of every channel tiling incl. split-K, and input gradients of strided convolutions on odd and even grids
    (1, 96, 0, 48, (5, 6, 7), 3, 2, True),
    (1, 192, 0, 96, (3, 4, 3), 3, 2, True),
    (2, 48, 0, 24, (4, 3, 18), 3, 2, True),
    (2, 48, 0, 48, (9, 10, 7), 3, 2, False),
    (1, 96, 0, 96, (7, 9, 5), 3, 2, False),
    (1, 192, 0, 192, (6, 4, 5), 3, 2, False),
    (1, 8, 0, 24, (10, 12, 8), 2, 2, False),        # conv_in (input channels padded 4 -> 8)
    (2, 48, 0, 24, (6, 7, 9), 1, 1, False),         # 1x1x1 residual conv
    (1, 192, 0, 384, (3, 4, 3), 3, 1, False),       # deep level: split-K over taps
    (1, 96, 96, 96, (4, 5, 4), 3, 1, False),
    # wide grids: the LDS halo-tile kernel (stride-1 3x3x3, 24-channel chunks), forward and input gradient
    (1, 24, 0, 24, (4, 13, 30), 3, 1, False),
    (1, 24, 24, 24, (3, 9, 33), 3, 1, False),
    (1, 48, 0, 48, (3, 10, 31), 3, 1, False),
    (2, 48, 0, 96, (2, 9, 30), 3, 1, False),
    (1, 96, 96, 192, (2, 8, 30), 3, 1, False),
    (1, 24, 0, 24, (3, 7, 65), 3, 1, False),        # the level-0 row width of BASELINE cfg4
    # channel counts that are not multiples of 48 (one weight-gradient launch per 48 x 48 block) on grids with >= 256 work
    # items per launch: every launch fills all 256 slabs of the workspace (round-2 advisor finding: the bound was 252)
    (4, 128, 0, 128, (64, 4, 4), 3, 1, False),
    (2, 128, 0, 128, (128, 4, 4), 1, 1, False),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward_backward_vs_float64(ob, case):
    B, Ca, Cb, Cout, sp, ks, stride, transposed = case
    torch.manual_seed(1)
    Cin = Ca + Cb
    x = rb(torch.randn(B, Cin, *sp))
    wshape = (Cin, Cout) + (ks,) * 3 if transposed else (Cout, Cin) + (ks,) * 3
    W = rb(torch.randn(wshape) / np.sqrt(Cin * ks ** 3))
    bias = torch.randn(Cout) * 0.1
    pad = 0 if ks == 1 else 1
    x64, W64 = x.double().requires_grad_(True), W.double().requires_grad_(True)
    b64 = bias.double().requires_grad_(True)
    if transposed:
        ref = F.conv_transpose3d(x64, W64, b64, stride=2, padding=1, output_padding=1)
    else:
        ref = F.conv3d(x64, W64, b64, stride=stride, padding=pad)
    xa = to_cl(x[:, :Ca]).requires_grad_(True)
    xb = to_cl(x[:, Ca:]).requires_grad_(True) if Cb else None
    Wd, bd = W.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
    y, mr = ob.ConvFn.apply(xa, xb, Wd, bd, ks, stride, transposed, True, 1e-5)
    assert tuple(y.shape) == (B,) + tuple(ref.shape[2:]) + (Cout,)
    assert rel_err(from_cl(y).numpy(), ref.detach().numpy()) < BF16_TOL
    # GroupNorm(1, C) statistics of the ROUNDED output
    yr = from_cl(y).double()
    mean = yr.mean(dim=(1, 2, 3, 4))
    rstd = 1.0 / torch.sqrt(yr.var(dim=(1, 2, 3, 4), unbiased=False) + 1e-5)
    assert float((mr[:, 0].cpu() - mean).abs().max()) < 1e-4 * float(yr.abs().max())
    assert rel_err(mr[:, 1].cpu().numpy(), rstd.numpy()) < 1e-4
    # backward with a bf16-representable cotangent
    cot = rb(torch.randn(ref.shape))
    gx64, gW64, gb64 = torch.autograd.grad((ref * cot.double()).sum(), [x64, W64, b64])
    ins = [xa] + ([xb] if Cb else []) + [Wd, bd]
    gs = torch.autograd.grad((y.float() * to_cl(cot).float()).sum(), ins)
    gxa = from_cl(gs[0])
    assert rel_err(gxa.numpy(), gx64[:, :Ca].numpy()) < BF16_TOL
    if Cb:
        assert rel_err(from_cl(gs[1]).numpy(), gx64[:, Ca:].numpy()) < BF16_TOL
    assert rel_err(gs[-2].cpu().numpy(), gW64.numpy()) < 1e-4          # fp32 accumulation of exact bf16 products
    assert rel_err(gs[-1].cpu().numpy(), gb64.numpy()) < 1e-4


@pytest.mark.parametrize('two', [False, True])
@pytest.mark.parametrize('act', ['elu', 'selu'])
def test_groupnorm_act_vs_float64(ob, two, act):
    from multimodal_3d_image_segmentation_amd import ops
    torch.manual_seed(2)
    B, C, sp = 2, 24, (5, 6, 7)
    aid = ops.act_id(act)
    fn = F.elu if act == 'elu' else F.selu

    def branch(seed):
        torch.manual_seed(seed)
        y = rb(torch.randn(B, C, *sp) * 1.5 + 0.3)
        return y, torch.randn(C) * 0.5 + 1.0, torch.randn(C) * 0.2
    (y1, g1, b1), (y2, g2, b2) = branch(3), branch(4)
    refs = []
    leaves = []
    for (y, g, b) in ((y1, g1, b1), (y2, g2, b2))[:2 if two else 1]:
        yy, gg, bb = y.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
        refs.append(fn(F.group_norm(yy, 1, gg, bb, 1e-5)))
        leaves += [yy, gg, bb]
    ref = sum(refs)

    def stats(y):
        yd = y.double()
        m = yd.mean(dim=(1, 2, 3, 4))
        r = 1.0 / torch.sqrt(yd.var(dim=(1, 2, 3, 4), unbiased=False) + 1e-5)
        return torch.stack([m, r], dim=1).float().cuda()
    d1 = [to_cl(y1).requires_grad_(True), stats(y1), g1.cuda().requires_grad_(True), b1.cuda().requires_grad_(True)]
    args = d1 + [aid]
    if two:
        d2 = [to_cl(y2).requires_grad_(True), stats(y2), g2.cuda().requires_grad_(True), b2.cuda().requires_grad_(True)]
        args += d2
    z = ob.GNActFn.apply(*args)
    assert rel_err(from_cl(z).numpy(), ref.detach().numpy()) < BF16_TOL
    cot = rb(torch.randn(ref.shape))
    gref = torch.autograd.grad((ref * cot.double()).sum(), leaves)
    wrt = [d1[0], d1[2], d1[3]] + ([d2[0], d2[2], d2[3]] if two else [])
    gs = torch.autograd.grad((z.float() * to_cl(cot).float()).sum(), wrt)
    for k, (a, b) in enumerate(zip(gs, gref)):
        if a.dtype == torch.bfloat16:
            assert rel_err(from_cl(a).numpy(), b.numpy()) < BF16_TOL, k
        else:
            assert rel_err(a.cpu().numpy(), b.numpy()) < 1e-4, k
    # the per-channel column sums of dy that ride on the gradient tensor (the producing convolution's bias gradient, taken from the
    # backward's own reductions instead of a pass over dy): against the float64 sums of the float64 gradient
    dy, dg, db, cs = ob.gn_bwd_raw(to_cl(cot), d1[0].detach(), d1[1], d1[2].detach(), d1[3].detach(), aid, colsum=True)
    want = gref[0].sum(dim=(0, 2, 3, 4)).numpy()
    assert rel_err(cs.cpu().numpy(), want) < 1e-4
    assert rel_err(from_cl(dy).double().sum(dim=(0, 2, 3, 4)).numpy(), want) < BF16_TOL     # what a pass over the bf16 dy gives


@pytest.mark.parametrize('shape', ['24+24->24', '24->24', 'branch'])
def test_pointwise_bf16_variants_vs_float64(shape):
    """bf16 matrix-core variants of the planar pointwise kernels (hno_pwconv_fwd / _bwd / _fwd_branch / _bwd_branch with
    HNO_ACT_BF16): operands rounded to bf16, fp32 accumulation, the convolution output rounded to bf16 -- against float64 on
    the same bf16-rounded operands.  Outputs within one bf16 rounding (4e-3), input gradients within 1.2e-2 (their operand,
    g * act'(y), is itself rounded to bf16), weight gradients (kept in fp32 tiles) within 1e-4 of the float64 ones."""
    import multimodal_3d_image_segmentation_amd as pkg
    ops = pkg.ops
    torch.manual_seed(3)
    B, V3 = 2, (5, 6, 7)
    xa = rb(torch.randn(B, 24, *V3)); xb = rb(torch.randn(B, 24, *V3))
    W = rb(torch.randn(24, 48 if shape != '24->24' else 24) * 0.2); bias = torch.randn(24) * 0.1
    if shape == 'branch':
        # y = selu(s + bf16(Wbr x + bbr)); out = selu(bf16(Wc [bf16(y) ; x] + bc))
        sop = torch.randn(B, 24, *V3); Wbr = rb(torch.randn(24, 24) * 0.2); bbr = torch.randn(24) * 0.1
        L = pkg._lib.lib()
        P, S = pkg._lib.ptr, pkg._lib.stream_ptr
        d = [t.cuda().contiguous() for t in (sop, xb, Wbr, bbr, W, bias)]
        y = torch.empty(B, 24, *V3, device='cuda'); out = torch.empty_like(y)
        pkg._lib.check(L.hno_pwconv_fwd_branch(P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(d[5]), P(y), P(out), B, 24, 24, 24,
                                               int(np.prod(V3)), ops.ACT_SELU | ops.ACT_BF16, S()), 'fwd_branch')
        br = rb(torch.einsum('oi,bidhw->bodhw', Wbr.double(), xb.double()).float() + bbr.view(1, -1, 1, 1, 1))
        y64 = F.selu(sop.double() + br.double())
        pre = torch.einsum('oi,bidhw->bodhw', W.double(), torch.cat([rb(y64.float()).double(), xb.double()], 1)) + bias.double().view(1, -1, 1, 1, 1)
        out64 = F.selu(rb(pre.float()).double())
        assert rel_err(y.cpu().numpy(), y64.numpy()) < BF16_TOL
        assert rel_err(out.cpu().numpy(), out64.numpy()) < 2 * BF16_TOL      # two roundings on the way
        return
    two = shape == '24+24->24'
    x64 = (torch.cat([xa, xb], 1) if two else xa).double().requires_grad_(True)
    W64, b64 = W.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = F.selu(torch.einsum('oi,bidhw->bodhw', W64, x64) + b64.view(1, -1, 1, 1, 1))
    cot = rb(torch.randn(ref.shape))
    gx64, gW64, gb64 = torch.autograd.grad((ref * cot.double()).sum(), [x64, W64, b64])
    xad = xa.cuda().requires_grad_(True)
    xbd = xb.cuda().requires_grad_(True) if two else None
    Wd, bd = W.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = ops.PwConvFn.apply(xad, xbd, Wd, bd, ops.ACT_SELU)
    assert y.dtype == torch.float32
    assert rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) < 2 * BF16_TOL     # pre-activation and output are both rounded
    y32 = ops.PwConvFn.apply(xad, xbd, Wd, bd, ops.ACT_SELU)
    assert rel_err(y32.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5 and not torch.equal(y, y32)   # the flag switched kernels
    gs = torch.autograd.grad((y * cot.cuda()).sum(), [xad] + ([xbd] if two else []) + [Wd, bd])
    assert rel_err(gs[0].cpu().numpy(), gx64[:, :24].numpy()) < 1.2e-2
    if two:
        assert rel_err(gs[1].cpu().numpy(), gx64[:, 24:].numpy()) < 1.2e-2
    # the weight gradient multiplies g * act'(y) (y here = the bf16-rounded output) by the inputs in fp32
    assert rel_err(gs[-2].cpu().numpy(), gW64.numpy()) < 1e-2
    assert rel_err(gs[-1].cpu().numpy(), gb64.numpy()) < 1e-2
