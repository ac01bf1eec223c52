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


# shapes: the two deepest V-Net levels of cfg4 (384 and 192 channels: few voxels, many workgroup rows per channel group) and a tensor with
# many voxels per workgroup
@pytest.mark.parametrize('shape', [(2, 24, (5, 6, 7)), (1, 384, (6, 7, 5)), (1, 192, (11, 13, 9)), (2, 24, (20, 24, 30))], ids=str)
@pytest.mark.parametrize('two', [False, True])
@pytest.mark.parametrize('act', ['elu', 'selu'])
def test_groupnorm_act_vs_float64(ob, two, act, shape):
    from multimodal_3d_image_segmentation_amd import ops
    torch.manual_seed(2)
    B, C, sp = shape
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
    if B * sp[0] * sp[1] * sp[2] <= 2000:      # what a pass over the bf16 dy gives (its rounding noise grows with the number of rows summed)
        assert rel_err(from_cl(dy).double().sum(dim=(0, 2, 3, 4)).numpy(), want) < BF16_TOL


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


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: bf16 activations IN MEMORY for the FNOSeg block chain (HNO_ACT_IO16; reference: what torch.autocast(bfloat16) makes of the
# block inputs / outputs, experiments/train_test.py:154-160 with nets/architectures.py:521-546).  Storing a tensor whose values are
# already bf16-representable as bf16 loses nothing, so every bf16-storage kernel is held BIT-EXACT to its fp32-storage twin.
def _padded(ops, B, C, N, seed, bf16_values=True):
    g = torch.Generator(device='cuda').manual_seed(seed)
    ld = ops._pad_ld(N ** 3)
    t = ops.act_empty(B, C, (N, N, N), 'cuda', ld)
    t.as_strided((B * C * ld,), (1,)).zero_()
    v = torch.randn((B, C, N, N, N), device='cuda', generator=g)
    t.copy_(v.bfloat16().float() if bf16_values else v)
    return t


def test_plane_transforms_with_bf16_planes_equal_the_fp32_kernels():
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd import ops
    from multimodal_3d_image_segmentation_amd._lib import lib, ptr, check, stream_ptr
    L = lib()
    B, C, N, modes = 1, 24, 65, (10, 14, 14)
    x = _padded(ops, B, C, N, 1)
    ld = ops.chan_stride(x)
    x16 = ops.to_bf16_layout(x, ld)
    assert ops.chan_stride16(x16) == ld and torch.equal(x16.float(), x)
    nws = L.hno_dht3_workspace_bytes(B * C, N, N, N, *modes) // 4
    ws32, ws16 = torch.zeros(nws, device='cuda'), torch.zeros(nws, device='cuda')
    check(L.hno_dht3_planes(ptr(x), ptr(ws32), B * C, N, N, N, *modes, ld, stream_ptr()), 'planes')
    check(L.hno_dht3_planes_b16(ptr(x16), ptr(ws16), B * C, N, N, N, *modes, ld, stream_ptr()), 'planes_b16')
    assert torch.equal(ws32, ws16)                    # same arithmetic on the same values: the intermediate is bit-identical
    # inverse: fp32 output rounded to bf16 afterwards == bf16 output of the kernel, with and without the fp32 residual
    add = _padded(ops, B, C, N, 2, bf16_values=False)
    for addend in (None, add):
        o32 = ops.act_empty(B, C, (N, N, N), 'cuda', ld)
        o16 = ops.act_empty16(B, C, (N, N, N), 'cuda', ld)
        check(L.hno_idht3_planes(ptr(ws32), ptr(addend), 0, ptr(o32), B * C, N, N, N, *modes, 0.5, ld, stream_ptr()), 'iplanes')
        check(L.hno_idht3_planes_b16(ptr(ws32), ptr(addend), 0, ptr(o16), B * C, N, N, N, *modes, 0.5, ld, stream_ptr()), 'iplanes_b16')
        assert torch.isfinite(o32).all() and float(o32.abs().max()) > 0
        assert torch.equal(o16, o32.bfloat16()), float((o16.float() - o32).abs().max())
        # the padding behind every channel's last voxel is written (zero): gradient padding must be exactly zero
        flat = o16.as_strided((B * C, ld), (ld, 1))
        assert float(flat[:, N ** 3:].float().abs().max()) == 0.0
    # sizes without a bf16 item kernel fail loudly (no silent fp32 fallback)
    y16 = torch.zeros(24 * 33 ** 3 + 64, device='cuda', dtype=torch.bfloat16)
    ws = torch.zeros(L.hno_dht3_workspace_bytes(24, 33, 33, 33, *modes) // 4, device='cuda')
    rc = L.hno_dht3_planes_b16(ptr(y16), ptr(ws), 24, 33, 33, 33, *modes, 0, stream_ptr())
    assert rc != 0 and b'65 x 65' in L.hno_last_error()


def test_block_tail_with_bf16_tensors_equals_fp32_storage():
    """hno_pwconv_fwd_branch / hno_pwconv_bwd_branch with HNO_ACT_IO16 against the same calls on fp32 tensors that hold the same
    (bf16-representable) values: y, out, p, g_x and all four parameter gradients bit for bit."""
    from multimodal_3d_image_segmentation_amd import ops
    from multimodal_3d_image_segmentation_amd._lib import lib, ptr, check, stream_ptr
    L = lib()
    B, C, N = 2, 24, 65
    torch.manual_seed(3)
    s, x = _padded(ops, B, C, N, 4, bf16_values=False), _padded(ops, B, C, N, 5)
    ld = ops.chan_stride(x)
    Wbr, bbr = torch.randn(24, 24, device='cuda') * 0.2, torch.randn(24, device='cuda') * 0.1
    W, b = torch.randn(24, 48, device='cuda') * 0.15, torch.randn(24, device='cuda') * 0.1
    act = ops.ACT_SELU
    x16 = ops.to_bf16_layout(x, ld)
    y32, o32, y16 = ops.act_like(x), ops.act_like(x), ops.act_like(x)
    o16 = ops.act_empty16(B, C, (N, N, N), 'cuda', ld)
    check(L.hno_pwconv_fwd_branch(ptr(s), ptr(x), ptr(Wbr), ptr(bbr), ptr(W), ptr(b), ptr(y32), ptr(o32), B, 24, 24, 24, ld, act | ops.ACT_BF16,
                                  stream_ptr()), 'fwd32')
    check(L.hno_pwconv_fwd_branch(ptr(s), ptr(x16), ptr(Wbr), ptr(bbr), ptr(W), ptr(b), ptr(y16), ptr(o16), B, 24, 24, 24, ld,
                                  act | ops.ACT_BF16 | ops.ACT_IO16, stream_ptr()), 'fwd16')
    assert torch.equal(y16, y32) and torch.equal(o16.float(), o32) and float(o32.abs().max()) > 0
    g = _padded(ops, B, C, N, 6)                     # the gradient of a bf16 tensor is bf16
    g.as_strided((B * C, ld), (ld, 1))[:, N ** 3:] = 0
    g16 = ops.to_bf16_layout(g, ld)
    r32 = ops.pwconv_bwd_branch_raw(g, o32, y32, x, W, Wbr, act, act, bf16=True)
    r16 = ops.pwconv_bwd_branch_raw(g16, o16, y16, x16, W, Wbr, act, act, bf16=True, io16=True)
    for name, a32, a16 in zip(('p', 'g_x', 'dW', 'db', 'dWbr', 'dbbr'), r32, r16):
        assert a16.dtype == torch.float32
        if name in ('dW', 'dWbr'):
            # the bf16-storage kernel forms the weight gradients from bf16-rounded operands on the bf16 matrix cores with fp32
            # accumulation -- what autocast does to the convolutions' weight gradients; held against float64 on the rounded operands
            continue
        assert torch.equal(a32, a16), name
    p32, gx32, dW32, _, dWbr32, _ = r32
    V = N ** 3
    flat = lambda t: t.as_strided((B, t.shape[1], V), (t.shape[1] * ld, ld, 1)).double()
    # g1 = g * act'(out); dW = sum_v g1 [y ; x]^T;  dWbr = sum_v p x^T  -- operands rounded to bf16 as the kernel rounds them
    o64, g64 = flat(o32), flat(g)
    g1 = (g64 * torch.where(o64 > 0, torch.full_like(o64, 1.0507009873554805), o64 + 1.0507009873554805 * 1.6732632423543772)).float().bfloat16().double()
    yx = torch.cat([flat(y32).float().bfloat16().double(), flat(x)], dim=1)
    dW_ref = torch.einsum('bov,biv->oi', g1, yx)
    dWbr_ref = torch.einsum('bov,biv->oi', flat(p32).float().bfloat16().double(), flat(x))
    assert rel_err(r16[2].cpu().numpy(), dW_ref.cpu().numpy()) < 1e-4
    assert rel_err(r16[4].cpu().numpy(), dWbr_ref.cpu().numpy()) < 1e-4
    assert rel_err(r16[2].cpu().numpy(), dW32.cpu().numpy()) < 1e-2 and rel_err(r16[4].cpu().numpy(), dWbr32.cpu().numpy()) < 1e-2
    # misuse fails loudly: bf16 tensors without bf16 arithmetic
    rc = L.hno_pwconv_fwd_branch(ptr(s), ptr(x16), ptr(Wbr), ptr(bbr), ptr(W), ptr(b), ptr(y16), ptr(o16), B, 24, 24, 24, ld, act | ops.ACT_IO16,
                                 stream_ptr())
    assert rc != 0


def test_fnoseg_chain_with_bf16_activations_in_memory(monkeypatch):
    """A 3-block FNOSeg (the BASELINE cfg3 block) under autocast with bf16 block inputs / outputs IN MEMORY (the default) against fp32
    storage (HNO_IO16=0): the forward is bit-identical (the stored values were bf16-representable already); the backward differs by the
    bf16 rounding of each block-input gradient -- what the reference's autocast run does to the same tensors."""
    import multimodal_3d_image_segmentation_amd as pkg
    from multimodal_3d_image_segmentation_amd import ops
    from multimodal_3d_image_segmentation_amd.nets import custom_losses
    torch.manual_seed(11)
    model = pkg.nets.NeuralOperatorSeg(4, 4, 24, 3, (10, 14, 14), 'Fourier').cuda()
    x = torch.randn(1, 4, 128, 128, 128, device='cuda')
    lab = pkg.ops.labels_prepare(torch.randint(0, 4, (1, 1, 128, 128, 128), device='cuda').float(), 4)
    seen = []
    real = ops.NOBlockFn.forward

    def spy(ctx, xin, *a):
        out = real(ctx, xin, *a)
        seen.append((xin.dtype, out.dtype))
        return out
    monkeypatch.setattr(ops.NOBlockFn, 'forward', staticmethod(spy))
    res = {}
    for io in ('1', '0'):
        monkeypatch.setenv('HNO_IO16', io)
        del seen[:]
        for p in model.parameters():
            p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = model(x)
            loss = custom_losses.PCCLoss()(y, lab)
        loss.backward()
        want = torch.bfloat16 if io == '1' else torch.float32
        assert seen == [(want, want)] * 3, seen
        res[io] = (y.detach().clone(), float(loss), torch.cat([p.grad.flatten() for p in model.parameters()]))
    assert torch.equal(res['1'][0], res['0'][0]) and res['1'][1] == res['0'][1]
    g1, g0 = res['1'][2], res['0'][2]
    err = float((g1 - g0).norm() / g0.norm())
    assert 0.0 < err < 2e-2, err
