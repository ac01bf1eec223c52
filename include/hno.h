/*
 * libhno -- C ABI of the MI355X (gfx950) spectral-operator segmentation kernels.
 *
 * The reference (IBM/multimodal-3d-image-segmentation) is pure PyTorch and has no FFI of
 * its own; every entry point below replaces the ATen op sequence issued by the cited
 * reference call site (file:line into the reference tree).  All pointers are DEVICE
 * pointers to contiguous fp32 NCDHW data unless stated otherwise, sizes are plain ints,
 * `stream` is a hipStream_t passed as void*.  Kernels are enqueued on `stream` and never
 * synchronise.  Return value: 0 on success, negative HNO_E* code otherwise (no C++
 * exception crosses the boundary); hno_last_error() returns a thread-local message.
 *
 * The library owns nothing except immutable per-(N, m, device) twiddle tables; all
 * buffers, including workspaces, are allocated and owned by the caller.
 */
#ifndef HNO_H_
#define HNO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HNO_OK 0
#define HNO_EINVAL (-1)   /* bad argument (size, mode count, null pointer)      */
#define HNO_ELIMIT (-2)   /* size outside the limits of the fused kernels        */
#define HNO_EHIP (-3)     /* a HIP runtime call failed                           */

#define HNO_ACT_NONE 0
#define HNO_ACT_SELU 1
#define HNO_ACT_ELU 2
#define HNO_ACT_SIGMOID 3 /* hno_act_fwd / hno_act_bwd only (output activation of the models); the fused conv / transform
                             epilogues take NONE, SELU or ELU */
/* ORed into the `act` argument of hno_pwconv_fwd / hno_pwconv_bwd / hno_pwconv_fwd_branch / hno_pwconv_bwd_branch: run the channel
 * contraction on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation) with the operands rounded to bf16 and the
 * convolution output rounded to bf16 -- what torch.autocast(bfloat16) makes of nn.Conv3d (reference experiments/train_test.py:154-160).
 * Tensors stay fp32 in memory.  Built for the 24- and 48-channel shapes of the BASELINE configurations (24 -> 24, 24 + 24 -> 24,
 * 24 -> 4, the fused block tail); other shapes ignore the flag and keep fp32 arithmetic.  The weight gradient keeps its fp32 tiles. */
#define HNO_ACT_BF16 0x1000
/* ORed in together with HNO_ACT_BF16 (round 6): the tensors torch.autocast makes bf16 ARE bf16 in memory -- a block's input and output
 * (and the output's gradient): hno_pwconv_fwd_branch takes x and writes out as bf16 (s, y fp32); hno_pwconv_bwd_branch takes gy, y, xb as
 * bf16 (xa and both gradient outputs fp32).  Rows are 2-byte elements with the SAME element stride V (V % 32 == 0: 64-byte aligned rows).
 * The transforms' counterparts: hno_dht3_planes_b16 (bf16 input), hno_idht3_planes_b16 (bf16 output). */
#define HNO_ACT_IO16 0x2000

int hno_version(void);
const char *hno_last_error(void);

/* ---------------------------------------------------------------- 3-D Hartley transform
 * hno_dht3_crop: out[bc, o0, o1, o2] = scale * sum_n x_eff[bc, n] * cas(phi(k(o), n))
 *   over the kept mode block [0..m) U [N-m..N) per axis ('[low | high]' order).
 *   x_eff = x                              if act_grad == HNO_ACT_NONE
 *         = x * act'(from saved output u)  otherwise (u = `x_act_out`, same shape as x)
 *   Replaces TransformCrop.forward (nets/hnosegxs.py:378-410: dhtn + 8 slices + 7 cats)
 *   with scale = 1/(N0 N1 N2), and the backward of PadInverse (+ fused activation grad)
 *   with scale = 1.  Modes must already be clamped (2 m_j <= N_j).
 * hno_pad_idht3: out[bc, n] = act(scale * sum_k z[bc, k] * cas(phi(k, n)) + addend[bc, n])
 *   Replaces PadInverse.forward (nets/hnosegxs.py:454-494: zeros + 7 cats + unscaled dhtn)
 *   with scale = 1, and the backward of TransformCrop with scale = 1/(N0 N1 N2) and the
 *   skip-connection gradient as `addend`.  `addend` may be NULL.
 * Both need a caller-provided workspace of hno_dht3_workspace_bytes(...) bytes.
 */
size_t hno_dht3_workspace_bytes(int BC, int N0, int N1, int N2, int m0, int m1, int m2);
int hno_dht3_crop(const float *x, const float *x_act_out, int act_grad, float *out, void *workspace,
                  int BC, int N0, int N1, int N2, int m0, int m1, int m2, float scale, void *stream);
int hno_pad_idht3(const float *z, const float *addend, int act, float *out, void *workspace,
                  int BC, int N0, int N1, int N2, int m0, int m1, int m2, float scale, void *stream);
/* hno_dht3_full: out[bc, k] = scale * sum_n x[bc, n] * cas(phi(k, n)) for EVERY frequency, natural order,
 *   any (even or odd) sizes with N0 <= 63; out has the shape of x.  Replaces dht.dhtn / dht2 / dht3
 *   (nets/dht.py:16-66; scale = 1/(N0 N1 N2) forward, 1 for is_inverse=True); N0 = 1 transforms each
 *   (N1, N2) plane (dht2).  The transform matrix is symmetric, so its backward is the same call.
 *   Workspace: hno_dht3_workspace_bytes(BC, N0, N1, N2, N0/2, N1/2, N2/2).
 * 2-D data: hno_dht3_crop / hno_pad_idht3 (and the rfft pair below) accept N0 = 1 with m0 = 0; the mode
 *   block then is (1, 2m1, 2m2) -- TransformCrop / PadInverse with ndim = 4 (nets/hnosegxs.py:361-376, :437-452).
 */
int hno_dht3_full(const float *x, float *out, void *workspace, int BC, int N0, int N1, int N2, float scale,
                  void *stream);

/* ---------------------------------------------------- mode-truncated real FFT (FNO path)
 * Same kernels as the Hartley pair with a different spectrum convention.  `spec` is the kept half
 * spectrum as REAL data (B, 2, C, 2m0, 2m1, m2): re plane then im plane per batch element, i.e. a
 * (B, 2C, ...) tensor whose channels are [re(0..C-1), im(0..C-1)], modes [low|high] on the first two
 * axes and [0, m2) on the last.
 * hno_rfft3_crop: spec = w(k2) * scale * sum_n x_eff[n] e^{-i phi(k, n)}.  k2_weights = 0 with scale
 *   1/(N0 N1 N2) replaces torch.fft.rfftn(norm='forward') + the 4 corner slices of
 *   FourierOperator._call3d (nets/fourier_operator.py:164-191); k2_weights = 1 (w = 1,2,2,...) with scale 1
 *   is the backward of hno_irfft3_pad.
 * hno_irfft3_pad: out = act(scale * sum_k w(k2) Re[spec[k] e^{+i phi(k, n)}] + addend).  k2_weights = 1,
 *   scale 1 replaces the zero padding + torch.fft.irfftn(norm='forward') (:195-209); k2_weights = 0 with
 *   scale 1/(N0 N1 N2) is the backward of hno_rfft3_crop.  Workspace: hno_dht3_workspace_bytes(B*C, ...).
 */
int hno_rfft3_crop(const float *x, const float *x_act_out, int act_grad, float *spec, void *workspace,
                   int B, int C, int N0, int N1, int N2, int m0, int m1, int m2, float scale, int k2_weights,
                   void *stream);
int hno_irfft3_pad(const float *spec, const float *addend, int act, float *out, void *workspace,
                   int B, int C, int N0, int N1, int N2, int m0, int m1, int m2, float scale, int k2_weights,
                   void *stream);
/* the same pair on channel-padded activations (see hno_dht3_crop_ld): ldbc = floats between consecutive (b, c) volumes */
int hno_rfft3_crop_ld(const float *x, const float *x_act_out, int act_grad, float *spec, void *workspace, int B, int C, int N0, int N1,
                      int N2, int m0, int m1, int m2, float scale, int k2_weights, long long ldbc, void *stream);
int hno_irfft3_pad_ld(const float *spec, const float *addend, int act, float *out, void *workspace, int B, int C, int N0, int N1, int N2,
                      int m0, int m1, int m2, float scale, int k2_weights, long long ldbc, void *stream);

/* ------------------------------------------------- shared-weight spectral channel mixing
 * L stacked layers  z_{l+1} = act(W_l z_l + residual * z_l)  on a (B, C, M) spectrum
 * (M = number of kept modes, contiguous).  W is (L, C, C) row-major [l][o][i].
 * zs receives the L layer outputs, each (B, C, M): zs[l] = z_{l+1}.
 * Replaces NeuralOperatorBlock.forward x n_XS (nets/hnosegxs.py:307-329) incl. the
 * einsum 'oi,bidhw->bodhw' of HartleyOperator._call3d_notransform
 * (nets/hartley_operator.py:287-292).
 * Backward: given g = dL/dz_L, the saved z_0 and zs, produces dL/dz_0 and WRITES
 * dL/dW into dW (L, C, C).  workspace: hno_specmix_bwd_workspace_bytes(B, C, M, L).
 */
size_t hno_specmix_bwd_workspace_bytes(int B, int C, int M, int L);
int hno_specmix_shared_fwd(const float *z0, const float *W, float *zs, int B, int C, int M, int L,
                           int residual, int act, void *stream);
int hno_specmix_shared_bwd(const float *g, const float *z0, const float *zs, const float *W, float *gz0,
                           float *dW, void *workspace, int B, int C, int M, int L, int residual, int act,
                           void *stream);
/* Same, with the layer weights given as a HOST array of L device pointers, each (C, C): the
 * form the model uses, because the reference keeps one Parameter per layer
 * (state-dict keys layers.{i}.conv_blocks.{j}.op.weight) and stacking them would cost a copy per
 * block and step.  dW stays one (L, C, C) block (a view per layer on the Python side).
 * L <= 64.  Up to C = 32 the whole stack is ONE kernel each way (weights and the 32-mode tile stay
 * in registers between layers); wider stacks run layer by layer through hno_pwconv. */
/* ------------------------------------------------------------ fused spectral middle of an HNO-XS block
 * TransformCrop's axis-D step + the n_XS frequency-domain layers + PadInverse's axis-D step in one kernel, between the two plane
 * transforms (nets/hnosegxs.py:378-410, 307-329, 454-494):
 *     hno_dht3_planes(x, ws)  ->  hno_spec_mid_fwd(ws, W, zs)  ->  hno_idht3_planes(ws, addend, act, out)
 * computes what hno_dht3_crop -> hno_specmix_layers_fwd -> hno_pad_idht3 compute.  The workspace (hno_dht3_workspace_bytes) is
 * transformed in place.  zs: (L + 1, B, C, 2 m0, 2 m1, 2 m2) = cropped spectrum z0 followed by the L layer outputs (what
 * the backward needs).  hno_spec_mid_supported says whether the fused kernels exist for a configuration (24 channels, N0 in
 * {33, 41, 49, 57, 65, 73, 81, 97} -- the working grids of 64^3 ... 192^3 inputs --, m0 = 10, m1, m2 <= 15); callers use the three-kernel path otherwise.  ldbc: stride (floats) between consecutive (b, c)
 * volumes of x / out / addend (0 = N0 N1 N2, contiguous). */
int hno_spec_mid_supported(int C, int N0, int m0, int m1, int m2, int L);
int hno_dht3_planes(const float *x, void *workspace, int BC, int N0, int N1, int N2, int m0, int m1, int m2, long long ldbc,
                    void *stream);
int hno_spec_mid_fwd(void *workspace, const float *const *W_layers, float *zs, int B, int C, int N0, int m0, int m1, int m2, int L,
                     int residual, int act, float scale, void *stream);
/* backward of the same chain: workspace = hno_dht3_planes of the block-output gradient on entry, operand of hno_idht3_planes (which
 * yields the block-input gradient) on return; zs as written by hno_spec_mid_fwd; dW (L, C, C); slab_workspace of slab_bytes >=
 * hno_spec_mid_bwd_workspace_bytes(B, C, m1, L) bytes (one slab of weight-gradient partial sums per workgroup: grows with the batch;
 * HNO_EINVAL if smaller); bit 8 of `residual` defers the slab reduction (hno_set_defer_reduce). */
size_t hno_spec_mid_bwd_workspace_bytes(int B, int C, int m1, int L);
int hno_spec_mid_bwd(void *workspace, const float *const *W_layers, const float *zs, float *dW, void *slab_workspace, size_t slab_bytes,
                     int B, int C, int N0, int m0, int m1, int m2, int L, int residual, int act, float scale, void *stream);
/* the same for the Fourier block (FNOSeg): D step of the rfft + crop -> complex channel mix as ONE real (2C, 2C) product on [re | im]
 * channels (hno_cmix_compose) -> zero pad + D step of the inverse, and its backward.  w_fwd / w_inv: c2r weights (1, 2, 2, ... over k2)
 * on the forward / inverse D step (forward pass: 0, 1; backward pass: 1, 0).  s0 (B, 2C, 2 m0, 2 m1, m2) = what hno_rfft3_crop returns
 * (written by _fwd, read by _bwd); dW2 (2C, 2C) for hno_cmix_split_grad; slab_workspace: slab_bytes >=
 * hno_spec_mid_fourier_bwd_workspace_bytes(B, C, m1). */
int hno_spec_mid_fourier_supported(int C, int N0, int m0, int m1, int m2);
int hno_spec_mid_fourier_fwd(void *workspace, const float *W2, float *s0, int B, int C, int N0, int m0, int m1, int m2, float scale,
                             int w_fwd, int w_inv, void *stream);
size_t hno_spec_mid_fourier_bwd_workspace_bytes(int B, int C, int m1);
int hno_spec_mid_fourier_bwd(void *workspace, const float *W2, const float *s0, float *dW2, void *slab_workspace, size_t slab_bytes, int B,
                             int C, int N0, int m0, int m1, int m2, float scale, int w_fwd, int w_inv, void *stream);
int hno_idht3_planes(const void *workspace, const float *addend, int act, float *out, int BC, int N0, int N1, int N2, int m0, int m1,
                     int m2, float scale, long long ldbc, void *stream);
/* hno_dht3_planes / hno_idht3_planes for activations that are bf16 in memory (HNO_ACT_IO16): x / out hold 2-byte elements at element
 * stride ldbc; the inverse's addend stays fp32 (and must sit at the same 4-element phase as out).  HNO_ELIMIT for plane sizes without a
 * bf16 item kernel (built: 65 x 65, the working grid of the BASELINE configurations). */
int hno_dht3_planes_b16(const void *x_bf16, void *workspace, int BC, int N0, int N1, int N2, int m0, int m1, int m2, long long ldbc,
                        void *stream);
int hno_idht3_planes_b16(const void *workspace, const float *addend, int act, void *out_bf16, int BC, int N0, int N1, int N2, int m0,
                         int m1, int m2, float scale, long long ldbc, void *stream);
/* hno_dht3_crop / hno_pad_idht3 on channel-padded activations: ldbc = stride (floats) between consecutive (b, c) volumes of x /
 * x_act_out, resp. out / addend (0 or N0 N1 N2 = contiguous; N0 N1 N2 <= ldbc < N0 N1 N2 + 64).
 * The inverse zeroes the padding of `out`.  hno_dht3_ld_supported: 1 when both directions take a padded stride for this geometry. */
int hno_dht3_ld_supported(int N0, int N1, int N2, int m0, int m1, int m2);
int hno_dht3_crop_ld(const float *x, const float *x_act_out, int act_grad, float *out, void *workspace, int BC, int N0, int N1, int N2,
                     int m0, int m1, int m2, float scale, long long ldbc, void *stream);
int hno_pad_idht3_ld(const float *z, const float *addend, int act, float *out, void *workspace, int BC, int N0, int N1, int N2,
                     int m0, int m1, int m2, float scale, long long ldbc, void *stream);

int hno_specmix_layers_fwd(const float *z0, const float *const *W_layers, float *zs, int B, int C, int M,
                           int L, int residual, int act, void *stream);
int hno_specmix_layers_bwd(const float *g, const float *z0, const float *zs, const float *const *W_layers,
                           float *gz0, float *dW, void *workspace, int B, int C, int M, int L, int residual,
                           int act, void *stream);

/* ------------------------------------------------------------ 1x1x1 convolution (+concat)
 * y[b, o, v] = act( sum_i W[o, i] * [xa ; xb][b, i, v] + bias[o] ),  W is (Cout, Ca+Cb).
 * xb may be NULL (Cb = 0); bias may be NULL.  Replaces torch.cat + ConvNormAct k=1
 * (nets/hnosegxs.py:274-275, 254-255, 153; nets/nets_utils.py:127-133) and the bias-free
 * conv_out (nets/hnosegxs.py:178).
 * Backward: gy is dL/dy, y the saved OUTPUT (for act'); writes gxa / gxb (either may be
 * NULL to skip) and WRITES dW (Cout, Ca+Cb) and dbias (Cout).
 * xa_act != HNO_ACT_NONE additionally multiplies gxa by act'(xa) where xa is itself the output of
 * that activation (fuses the SELU backward of PadInverse into this kernel: HNOXSBlock :267-275).
 * accumulate_gx: bit 0 adds the input gradient to the values already in gxa, bit 1 likewise for gxb (3 = both):
 * gradient accumulation for a tensor with several consumers is fused into the store.
 * Weight gradients are reduced through per-block slabs in `workspace`
 * (hno_pwconv_bwd_workspace_bytes) in a fixed order: no float atomics, reproducible bit for bit.
 */
size_t hno_pwconv_bwd_workspace_bytes(int Cin, int Cout);
int hno_pwconv_fwd(const float *xa, int Ca, const float *xb, int Cb, const float *W, const float *bias,
                   float *y, int B, int Cout, long long V, int act, void *stream);
int hno_pwconv_bwd(const float *gy, const float *y, const float *xa, int Ca, const float *xb, int Cb,
                   const float *W, float *gxa, float *gxb, float *dW, float *dbias, void *workspace,
                   int B, int Cout, long long V, int act, int xa_act, int accumulate_gx, void *stream);

/* hno_pwconv_bwd with a FUSED CONV BRANCH (the FNOSeg / HNOSeg block, nets/architectures.py:521-546): the conv's first input
 * xa = act(s + Wbr xb + bbr) was itself produced from xb by a second 1x1x1 conv (conv_branch) plus the operator output s.
 * One pass over the voxels computes, with p = (W^T g)[xa rows] * act'(xa):
 *   p_out (B, Ca, V) = p                     -- the gradient of the pre-activation sum: feeds the operator's backward
 *   gxb   (B, Cb, V) = (W^T g)[xb rows] + Wbr^T p
 *   dflat = [ dW (Cout x (Ca+Cb)) | dbias (Cout) | dWbr (Ca x Cb) | dbbr (Ca) ]   (one flat buffer)
 * where g = gy * act'(y).  xa_act names the activation of xa (required).  Built for Ca = Cb = Cout = 24 (HNO_ELIMIT
 * otherwise: call hno_pwconv_bwd twice).  Workspace: hno_pwconv_bwd_branch_workspace_bytes. */
/* forward twin: y = act(s + Wbr xb + bbr); out = act(W [y ; xb] + bias) in one pass (y is written because the backward
 * needs it).  s: the operator output (B, Ca, V).  Same shape limit. */
int hno_pwconv_fwd_branch(const float *s, const float *xb, const float *Wbr, const float *bbr, const float *W,
                          const float *bias, float *y, float *out, int B, int Ca, int Cb, int Cout, long long V,
                          int act, void *stream);
/* Two chained pointwise layers in one pass (round 4): xi = act(Wc [u ; t] + bc), xn = act(Wm [xi ; k] + bm) -- the conv_concat that
 * ends HNO-XS block i and the mapping_conv over cat[block output, U-Net skip] that opens decoder block i + 1 (nets/hnosegxs.py:161-162,
 * 253-255, 274-275).  xi is written for the backward but never read back: 5 activation streams instead of 6.  (B, 24, V) tensors
 * (V may be the channel stride of channel-padded activations), Wc / Wm (24, 48); other widths: HNO_ELIMIT (callers run the two layers). */
int hno_pwconv_fwd_chain_supported(int C, int C2, int has_k);
/* C2 = 24 with k: the next block's mapping_conv (act2 = act);  C2 = 4, k = bm = NULL: the model's conv_out (nets/hnosegxs.py:178, 24 ->
 * out_channels, no bias, act2 none) behind the LAST block's conv_concat; xn (B, C2, V), Wm (C2, 48 or 24).  xi NULL: the first layer's
 * output is not stored (inference: only the backward reads it) */
int hno_pwconv_fwd_chain(const float *u, const float *t, const float *k, const float *Wc, const float *bc, const float *Wm,
                         const float *bm, float *xi, float *xn, int B, int C, int C2, long long V, int act, int act2, void *stream);
/* backward of the pair in one pass: gn = gradient of xn -> gu (times xa_act'(u): u is the output of that activation), gt, gk and
 * grads = [dWc (24, 48) | dbc (24) | dWm (C2, 48 | 24) | dbm (C2; not written in the C2 = 4 form)] in one flat buffer (the parameters' order in the model).  9 activation streams instead of 12 (5 + 2 x 4 ch
 * instead of 8 for the conv_out form): the gradient between the two layers never reaches memory.  k / gk NULL for the conv_out form.
 * workspace: hno_pwconv_bwd_chain_workspace_bytes(C); bit 8 of xa_act defers the slab reduction (hno_set_defer_reduce). */
size_t hno_pwconv_bwd_chain_workspace_bytes(int C);
int hno_pwconv_bwd_chain(const float *gn, const float *xn, const float *xi, const float *k, const float *u, const float *t,
                         const float *Wm, const float *Wc, float *gu, float *gt, float *gk, float *grads, void *workspace, int B, int C,
                         int C2, long long V, int act, int act2, int xa_act, void *stream);
size_t hno_pwconv_bwd_branch_workspace_bytes(int Ca, int Cb, int Cout);
int hno_pwconv_bwd_branch(const float *gy, const float *y, const float *xa, int Ca, const float *xb, int Cb,
                          const float *W, const float *Wbr, float *p_out, float *gxb, float *dflat, void *workspace,
                          int B, int Cout, long long V, int act, int xa_act, void *stream);

/* Complex shared-weight mix of the Fourier operator (nets/fourier_operator.py:164-172, complex 'oi,bi...->bo...') as ONE
 * real pointwise conv on the [re | im] channel layout: hno_cmix_compose builds W2 = [[Wr, -Wi], [Wi, Wr]] (2Co x 2Ci) for
 * hno_pwconv_fwd / hno_pwconv_bwd; hno_cmix_split_grad turns dW2 into dWr = dW2[re,re] + dW2[im,im], dWi = dW2[im,re] -
 * dW2[re,im]. */
int hno_cmix_compose(const float *w_real, const float *w_imag, float *w2, int Co, int Ci, void *stream);
/* the same for n <= 64 pairs of equal shape in one launch: w2_all (n, 2Co, 2Ci) (all Fourier blocks of a model, once per forward pass) */
int hno_cmix_compose_multi(const float *const *w_real, const float *const *w_imag, float *w2_all, int n, int Co, int Ci, void *stream);
int hno_cmix_split_grad(const float *dw2, float *dw_real, float *dw_imag, int Co, int Ci, void *stream);
/* defer != 0 and deferred slab reductions on (hno_set_defer_reduce): the split is recorded and launched by hno_flush_reduces behind the
 * reduction that writes dw2 (hno_spec_mid_fourier_bwd with bit 8 of w_fwd set) -- one kernel for all blocks of a backward pass */
int hno_cmix_split_grad_ex(const float *dw2, float *dw_real, float *dw_imag, int Co, int Ci, int defer, void *stream);

/* ------------------------------------------- strided 2x2x2 'resize' convolution (conv_in)
 * Conv3d(Cin -> Cout, kernel 2, stride 2, padding 1) + bias + act: (B,Cin,D,H,W) ->
 * (B,Cout,D/2+1,H/2+1,W/2+1).  Replaces ConvNormAct(kernel_size=2, stride=2)
 * (nets/hnosegxs.py:102-104,151; nets/nets_utils.py:156-163).  W is (Cout,Cin,2,2,2).
 * Backward writes dW / dbias (workspace = hno_pwconv_bwd_workspace_bytes(Cin*8, Cout))
 * and, if gx != NULL, the input gradient.
 * ldy: channel stride (floats) of y / gy (0 = the voxel count; or padded up to a multiple of 32, "channel-padded" activations:
 * every channel row then starts on a 128-byte boundary, which the pointwise backward needs to run at the memory rate -- the forward
 * zeroes the padding).
 */
int hno_conv_k2s2_fwd(const float *x, const float *W, const float *bias, float *y, int B, int Cin, int Cout,
                      int D, int H, int Wd, int act, long long ldy, void *stream);
int hno_conv_k2s2_bwd(const float *gy, const float *y, const float *x, const float *W, float *gx, float *dW,
                      float *dbias, void *workspace, int B, int Cin, int Cout, int D, int H, int Wd, int act,
                      long long ldy, void *stream);

/* Round 4: the stem of HNOSeg-XS in one pass each way -- conv_in = Conv3d(k 2, s 2, p 1) + bias + act followed by conv1 = Conv3d(k 1) +
 * bias + act1 (nets/hnosegxs.py:102-108, 151-152).  Forward: y1 only (conv_in's output stays in the matrix-core accumulators and is not
 * written).  Backward: recomputes conv_in's output from the image tile it reads anyway; grads = [dW_in (C0 x Cin x 8) | db_in (C0) |
 * dW1 (C1 x C0) | db1 (C1)]; workspace of hno_conv_k2s2_chain_bwd_workspace_bytes; bit 8 of act1 defers the slab reduction.
 * Cin <= 4, C0, C1 <= 32 (hno_conv_k2s2_chain_supported). */
int hno_conv_k2s2_chain_supported(int Cin, int C0, int C1);
size_t hno_conv_k2s2_chain_bwd_workspace_bytes(int Cin, int C0, int C1);
int hno_conv_k2s2_chain_fwd(const float *x, const float *W, const float *bias, const float *W1, const float *bias1, float *y1, int B, int Cin,
                            int C0, int C1, int D, int H, int Wd, int act, int act1, long long ldy, void *stream);
int hno_conv_k2s2_chain_bwd(const float *gy1, const float *y1, const float *x, const float *W, const float *bias, const float *W1, float *grads,
                            void *workspace, int B, int Cin, int C0, int C1, int D, int H, int Wd, int act, int act1, long long ldy,
                            void *stream);

/* ------------------------------------------------- output head: upsample + channel softmax
 * probs[b, c, :] = softmax_c( trilinear(logits_lr[b, c], size=(D,H,W), align_corners=False) )
 * logits_lr is (B, K, d, h, w).  Together with hno_pwconv_fwd(conv_out) at LOW resolution
 * this replaces F.interpolate(24 ch) -> conv_out -> softmax (nets/hnosegxs.py:174-180);
 * the 1x1x1 conv commutes with the (linear, per-channel) interpolation.
 * Backward: g_lr = trilinear^T( softmax'(probs, g_probs) ).  With a workspace of
 * hno_upsoftmax_bwd_workspace_bytes() the adjoint runs separably (W and H axes folded while the
 * gradient streams through once, then the D axis); with workspace NULL, or for shapes the separable
 * kernels do not cover (the size query returns 0), a gather kernel is used.
 */
int hno_upsoftmax_fwd(const float *logits_lr, float *probs, int B, int K, int d, int h, int w,
                      int D, int H, int W, int softmax, void *stream);
/* Inference head: labels[b, z, y, x] = argmax_c trilinear(logits_lr[b, c]) as uint8 -- the prediction of
 * testing() (experiments/train_test.py:398-408: model(x) -> cpu -> argmax(1) -> uint8) without forming or moving
 * the probabilities. */
int hno_up_argmax(const float *logits_lr, unsigned char *labels, int B, int K, int d, int h, int w, int D, int H,
                  int W, void *stream);
size_t hno_upsoftmax_bwd_workspace_bytes(int B, int K, int d, int h, int w, int D, int H, int W);
int hno_upsoftmax_bwd(const float *g_probs, const float *probs, float *g_lr, void *workspace, int B, int K, int d,
                      int h, int w, int D, int H, int W, int softmax, void *stream);
/* the head on a channel-padded low-resolution tensor (ldlr = floats between consecutive (b, k) volumes of logits_lr / g_lr, 0 = d h w;
 * the backward zeroes the padding of g_lr and needs the separable form: workspace > 0) */
int hno_upsoftmax_fwd_ld(const float *logits_lr, float *probs, int B, int K, int d, int h, int w, int D, int H, int W, int softmax,
                         long long ldlr, void *stream);
int hno_up_argmax_ld(const float *logits_lr, unsigned char *labels, int B, int K, int d, int h, int w, int D, int H, int W, long long ldlr,
                     void *stream);
int hno_upsoftmax_bwd_ld(const float *g_probs, const float *probs, float *g_lr, void *workspace, int B, int K, int d, int h, int w, int D,
                         int H, int W, int softmax, long long ldlr, void *stream);

/* ------------------------------------------------------------------ V-Net-DS building blocks
 * 3x3x3 convolutions as implicit GEMMs on the fp32 matrix cores (reference: ConvNormAct /
 * ConvTransposeNormAct with kernel_size 3, nets/nets_utils.py:136-211, used by VNetDS
 * nets/architectures.py:105-155).  hno_conv3d_k3 mode: 0 = Conv3d forward (W (Cout,Cin,3,3,3), stride 1 or 2,
 * padding 1); 1 = its input gradient (x = dL/dy with the conv's OUTPUT dims as Di.., y = dL/dx); 2 =
 * ConvTranspose3d forward (Wt (Cin,Cout,3,3,3), stride 2, padding 1, output_padding 1); 3 = its input
 * gradient.  Cin / Cout always name the ORIGINAL operator's channels.  workspace: at least
 * hno_conv3d_k3_workspace_bytes(Cin, Cout, 0) (weight re-layout); with hno_conv3d_k3_fwd_workspace_bytes(mode, B, ...,
 * Do, Ho, Wo) bytes (Do.. = dims of THIS call's output y) small grids are additionally split over the 27 taps so that the
 * deep V-Net levels fill the chip.  hno_conv3d_k3_wgrad writes dW (same layout as the weight);
 * Dx.. are the operator's input dims, Dg.. its output dims; workspace: ..._workspace_bytes(Cin, Cout, 1). */
size_t hno_conv3d_k3_workspace_bytes(int Cin, int Cout, int for_wgrad);
size_t hno_conv3d_k3_fwd_workspace_bytes(int mode, int B, int Cin, int Cout, int Do, int Ho, int Wo);
int hno_conv3d_k3(const float *x, const float *W, const float *bias, float *y, void *workspace, size_t workspace_bytes,
                  int mode, int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int stride, int pad,
                  int act, void *stream);
int hno_conv3d_k3_wgrad(const float *g, const float *x, float *dW, void *workspace, int transposed, int B, int Cin, int Cout,
                        int Dx, int Hx, int Wx, int Dg, int Hg, int Wg, int stride, int pad, void *stream);
/* Convolutions of V-Net-DS with ANY odd kernel size (the reference's `kernel_size` constructor argument, nets/architectures.py:55-70):
 * direct (untuned) kernels behind the same mode numbers as hno_conv3d_k3 -- 0 Conv3d forward (W (Cout, Cin, k, k, k), stride 1 'same' or
 * stride 2 padding k / 2), 1 its input gradient, 2 ConvTranspose3d forward (W (Cin, Cout, k, k, k), stride 2, padding k / 2,
 * output_padding 1), 3 its input gradient; (Di, Hi, Wi) / (Do, Ho, Wo) are the sizes of the tensor read / written.  hno_convk_wgrad
 * writes dW in the weight's layout; bias gradients are hno_channel_sum.  k = 3 runs the implicit-GEMM kernels above. */
int hno_convk(const float *x, const float *W, const float *bias, float *y, int mode, int B, int Cin, int Cout, int Di, int Hi, int Wi,
              int Do, int Ho, int Wo, int k, int stride, int pad, void *stream);
int hno_convk_wgrad(const float *g, const float *x, float *dW, int transposed, int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do,
                    int Ho, int Wo, int k, int stride, int pad, void *stream);
/* GroupNorm(1, C) + activation (nn.GroupNorm(1, C) after every V-Net conv, nets/nets_utils.py:165-170).
 * fwd saves mean_rstd (B,2); stats_ws: 2*B doubles.  bwd: sums_ws 2*B*C doubles, coef_ws 2*B floats. */
int hno_groupnorm1_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean_rstd, double *stats_ws,
                       int B, int C, long long V, float eps, int act, void *stream);
int hno_groupnorm1_bwd(const float *g, const float *y, const float *x, const float *mean_rstd, const float *gamma, float *gx,
                       float *dgamma, float *dbeta, double *sums_ws, float *coef_ws, int B, int C, long long V, int act,
                       void *stream);
/* nearest-neighbour resampling (F.interpolate default mode; `upsampling`, nets/architectures.py:638-653):
 * adjoint = 0: dst (BC,D,H,W) = src (BC,d,h,w) upsampled (accumulate: dst += ...); adjoint = 1: src is the
 * high-resolution gradient (BC,D,H,W), dst the low-resolution one (BC,d,h,w). */
int hno_nearest3d(const float *src, float *dst, int BC, int d, int h, int w, int D, int H, int W, int adjoint, int accumulate,
                  void *stream);
/* out[c] = sum_{b,v} g[b][c][v] (bias gradient of the 3x3x3 convolutions) */
size_t hno_channel_sum_workspace_bytes(int C);
int hno_channel_sum(const float *g, float *out, void *workspace, int B, int C, long long V, void *stream);

/* ------------------------------------------------------- bf16 matrix-core path (autocast)
 * The reference trains V-Net-DS / FNOSeg under torch.autocast(bfloat16) + GradScaler when `use_autocast` is set
 * (experiments/train_test.py:79,154-168): nn.Conv3d / nn.ConvTranspose3d / einsum take bf16 operands, GroupNorm and the
 * parameters stay fp32.  These entry points are that path: activations are CHANNELS-LAST bf16 (b, d, h, w, c), channel
 * counts multiples of 8; accumulation is fp32 (v_mfma_f32_32x32x16_bf16 / _16x16x32_bf16); parameters, their gradients and
 * the GroupNorm statistics are fp32.  `void *` tensors are bf16.  Replaces nets/nets_utils.py:127-211 (ConvNormAct /
 * ConvTransposeNormAct forward + backward) for nets/architectures.py:26-252. */
size_t hno_cb_packed_weight_bytes(int Cin, int Cout, int ks);
/* role 0: conv forward W[Cout][Cin][ks^3]; 1: conv input gradient (same tensor); 2: ConvTranspose forward Wt[Cin][Cout][ks^3];
 * 3: ConvTranspose input gradient.  dst: bf16 [tap * Ci/8 + c8][round_up(Co, 32)][8] of the GEMM's input (Ci) / output (Co) channels */
int hno_cb_pack_weights(const float *W, void *dst, int role, int Cin, int Cout, int ks, void *stream);
/* roles 0 + 1 (transposed = 0) or 2 + 3 (transposed = 1) in ONE launch: the forward operand and the input-gradient operand */
int hno_cb_pack_weights_both(const float *W, void *dst_fwd, void *dst_bwd, int transposed, int Cin, int Cout, int ks, void *stream);
/* all layers of a model in one launch: `table_dev` is an (nrows, 16) int64 device array whose rows were filled on the host by
 * hno_cb_pack_table_row (parameter pointer, the two destination buffers, layer shape) */
int hno_cb_pack_table_row(long long *row, const float *W, void *dst_fwd, void *dst_bwd, int transposed, int Cin, int Cout, int ks);
/* total_chunks = sum over rows of ceil((row[10] * row[9] + row[15] * row[14]) * 8 / 2048): one workgroup per 2048 packed elements */
long long hno_cb_pack_row_chunks(const long long *row);   /* workgroups of hno_cb_pack_weights_multi for this (host) row; total_chunks = their sum.
                                                             The destination buffers must be zeroed once by the caller: padding is not rewritten. */
int hno_cb_pack_weights_multi(const void *table_dev, int nrows, long long total_chunks, void *stream);
size_t hno_cb_conv_workspace_bytes(int B, int Cin, int Cout, int Do, int Ho, int Wo, int ks);
/* y = conv([xa ; xb]) + bias as a gather GEMM.  mode 0: in = stride * out - pad + tap (Conv3d forward, ConvTranspose3d input
 * gradient); mode 1: in = (out + pad - tap) / stride where divisible (ConvTranspose3d forward, Conv3d input gradient).
 * mean_rstd (B, 2) != NULL: GroupNorm(1, Cout) statistics of the bf16-rounded output come out of the same pass.
 * nstat_out != NULL ("lazy statistics", round 4): mean_rstd must hold hno_cb_conv_stats_floats() floats; the per-workgroup (sum, sum of
 * squares) partials are left behind its (B, 2) slot and their count per sample is returned instead of launching the one-workgroup
 * finalize kernel; hno_cb_gn_apply called with that count finishes them (and stores mean / rstd into the slot for the backward). */
size_t hno_cb_conv_stats_floats(int B, int Cout, int Do, int Ho, int Wo);
int hno_cb_conv(const void *xa, int Ca, const void *xb, int Cb, const void *wpacked, const float *bias, void *y, float *mean_rstd,
                float eps, void *workspace, size_t workspace_bytes, int mode, int B, int Cout, int Di, int Hi, int Wi, int Do, int Ho,
                int Wo, int ks, int stride, int pad, int *nstat_out, void *stream);
/* hno_cb_conv with its output channels written to TWO tensors (round 5): channels [0, c_split) to y (B, Do, Ho, Wo, c_split), the others
 * to y2 (B, Do, Ho, Wo, Cout - c_split); c_split a multiple of 8.  No statistics.  The input gradient of a two-input convolution -- the
 * reference's Conv3d over torch.cat([x, skip]) in the V-Net decoder (nets/architectures.py:226-252) -- is the gradient of both inputs;
 * autograd's split of the concatenated gradient (two strided copies per convolution) never happens. */
int hno_cb_conv_split(const void *xa, int Ca, const void *xb, int Cb, const void *wpacked, const float *bias, void *y, void *y2, int c_split,
                      void *workspace, size_t workspace_bytes, int mode, int B, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                      int ks, int stride, int pad, void *stream);

/* upper bound of the slab workspace hno_cb_wgrad writes for a layer with these (total input, output) channel counts */
size_t hno_cb_wgrad_workspace_bytes(int Cin, int Cout, int ks);
/* dW (fp32, the parameter's own layout) of a Conv3d (transposed = 0: g on the output grid (Dg, Hg, Wg), x = [xa ; xb] on the
 * input grid (Dx, Hx, Wx)) or of a ConvTranspose3d (transposed = 1: x is ITS input on the small grid, g its output gradient). */
int hno_cb_wgrad(const void *g, int Cg, const void *xa, int Ca, const void *xb, int Cb, float *dW, void *workspace,
                 size_t workspace_bytes, int transposed, int B, int Dx, int Hx, int Wx, int Dg, int Hg, int Wg, int ks, int stride,
                 int pad, void *stream);
/* z = act(GroupNorm(1, C)(y1)) [+ act(GroupNorm(1, C)(y2))]: the residual sum of a V-Net section fused (architectures.py:205-224).
 * nstat1 / nstat2 > 0: mr holds the producing hno_cb_conv's lazy statistics (partials behind the (B, 2) slot): finished here with
 * `eps`, (mean, rstd) stored into the slot; 0: mr already holds (mean, rstd). */
int hno_cb_gn_apply(const void *y1, const float *mr1, const float *gamma1, const float *beta1, const void *y2, const float *mr2,
                    const float *gamma2, const float *beta2, void *z, int B, int C, long long V, int act, int nstat1, int nstat2,
                    float eps, void *stream);
size_t hno_cb_gn_bwd_workspace_bytes(int B, int C);
/* dy_colsum (C floats, may be NULL): sum over samples and voxels of dy per channel = the bias gradient of the convolution that produced
 * y, obtained from the per-channel reductions the backward already makes (no pass over dy: see cb_gn_bwd_apply_kernel). */
int hno_cb_gn_bwd(const void *dz, const void *y, const float *mr, const float *gamma, const float *beta, void *dy, float *dgamma,
                  float *dbeta, float *dy_colsum, void *workspace, int B, int C, long long V, int act, int accumulate, void *stream);
/* fp32 NCDHW (C channels) -> bf16 channels-last with CP >= C channels (pad channels zero), and back */
int hno_cb_pack_input(const float *x, void *y, int B, int C, int CP, long long V, void *stream);
int hno_cb_unpack(const void *x, float *y, int B, int C, int CP, long long V, void *stream);
/* out[c] = sum over rows of a (rows, C) bf16 tensor (bias gradients) */
size_t hno_cb_colsum_workspace_bytes(int C);
int hno_cb_colsum(const void *g, float *out, void *workspace, int C, long long rows, void *stream);

/* ------------------------------------------------------- per-mode ('individual') spectral weights
 * Hartley (fourier = 0): y(k) = 1/2 [W(k)(x(k) + xr(k)) + W(-k)(x(k) - xr(k))] with xr = the frequency-
 *   reversed copy of x supplied by the caller (on the cropped grid or taken from the full spectrum) and
 *   W(-k) the reversal of W on its own (d0,d1,d2) = (2m0,2m1,2m2) grid.  x, xr (B,Ci,M); w (Co,Ci,M).
 *   Replaces hartley_conv + get_reverse (nets/hartley_operator.py:302-333).
 * Fourier (fourier = 1): complex y(k) = W(k) x(k); x / y hold [re | im] planes (B,2,C,M), w / wi the real
 *   and imaginary weights (Co,Ci,M).  Replaces einsum('oidhw,bidhw->bodhw') (nets/fourier_operator.py:174-191).
 * Backward writes gx (and gxr for Hartley), dw (and dwi for Fourier).  B <= 8 per call. */
int hno_permode_fwd(const float *x, const float *xr, const float *w, const float *wi, float *y, int B, int Ci, int Co,
                    int M, int d0, int d1, int d2, int fourier, void *stream);
int hno_permode_bwd(const float *g, const float *x, const float *xr, const float *w, const float *wi, float *gx, float *gxr,
                    float *dw, float *dwi, int B, int Ci, int Co, int M, int d0, int d1, int d2, int fourier,
                    void *stream);

/* ------------------------------------------------------------------------ batched GEMM
 * C[b] = alpha * op(A[b]) * op(B[b]) for b < batch; contiguous row-major operands, op = transpose when
 * the flag is set (A is M x K, or K x M when transA; B is K x N, or N x K when transB; C is M x N).
 * Replaces the two einsums of the Hartley attention (nets/hartley_mha.py:196-201) and their backward. */
int hno_bmm(const float *A, const float *B, float *C, int batch, int M, int N, int K, int transA, int transB,
            float alpha, void *stream);

/* Fused Hartley attention: out[c][q] = sum_k v[c][k] act(alpha sum_c' q[c'][q] k[c'][k]) per (batch, head) = BZ pairs, the
 * T x T attention matrix never materialised (forward and backward recompute its 32 x 32 tiles on the fp32 matrix cores).
 * q, k (BZ, Ck, T); v, out, dout (BZ, Cv, T); Ck, Cv <= 128 (hno_hmha_supported).  Replaces nets/hartley_mha.py:196-201
 * (einsum 'bzcq,bzck->bzqk' / sqrt(C), attention activation -- SELU, not softmax --, einsum 'bzqk,bzck->bzcq') and its backward. */
int hno_hmha_supported(int Ck, int Cv);
/* workspace (hno_hmha_workspace_bytes; round 4): the streamed side is split over several workgroups that share every fetched tile
 * among four owner waves; each split writes a partial result there and a small kernel adds them in a fixed order.  workspace = NULL
 * (or too small) runs the round-2 kernels, which need none. */
size_t hno_hmha_workspace_bytes(int BZ, int Ck, int Cv, int T);
int hno_hmha_fwd(const float *q, const float *k, const float *v, float *out, void *workspace, size_t workspace_bytes, int BZ, int Ck,
                 int Cv, int T, float alpha, int act, void *stream);
int hno_hmha_bwd(const float *q, const float *k, const float *v, const float *dout, float *dq, float *dk, float *dv, void *workspace,
                 size_t workspace_bytes, int BZ, int Ck, int Cv, int T, float alpha, int act, void *stream);

/* Round 4b: grouping3d / ungrouping3d (nets/hartley_mha.py:473-524) as one permutation kernel.  full: (B, C0 + C1 + C2, d, h, w), the
 * stacked q / k / v projections (or one tensor: C1 = C2 = 0); p_s: (B, C_s pd ph pw, (d / pd)(h / ph)(w / pw)) contiguous, NULL = skipped
 * (inverse: that channel range of `full` is zero-filled).  inverse = 0: full -> parts (grouping3d); 1: parts -> full (ungrouping3d). */
int hno_patch_group3(float *full, float *p0, float *p1, float *p2, int B, int C0, int C1, int C2, int d, int h, int w, int pd, int ph,
                     int pw, int inverse, void *stream);

/* Round 4b: the fused attention with the stream splits' partial results left unsummed -- hno_hmha_nsplit(BZ, T) slices of (BZ, C, T) floats
 * per output -- and the permutation that adds them in slice order while it ungroups (parts -> full only).  Saves the partial-sum launches
 * (4 per attention block and step). */
int hno_hmha_nsplit(int BZ, int T);
int hno_hmha_nsplit_bwd(int BZ, int Ck, int Cv, int T);   /* slices of hno_hmha_bwd_parts' outputs (<= hno_hmha_nsplit: round 6) */
int hno_hmha_parts_supported(int Ck, int Cv, int act);
int hno_hmha_fwd_parts(const float *q, const float *k, const float *v, float *out_parts, int BZ, int Ck, int Cv, int T, float alpha, int act,
                       void *stream);
int hno_hmha_bwd_parts(const float *q, const float *k, const float *v, const float *dout, float *dq_parts, float *dk_parts, float *dv_parts,
                       int BZ, int Ck, int Cv, int T, float alpha, int act, void *stream);
int hno_patch_group3_sum(float *full, const float *p0, const float *p1, const float *p2, int nsum, int B, int C0, int C1, int C2, int d, int h,
                         int w, int pd, int ph, int pw, void *stream);

/* ------------------------------------------------------------------- elementwise helpers
 * y = act(x) ; gx = g * act'(y) (y = saved output) ; out = a + b.  Used where the reference applies an
 * activation or a residual add that no neighbouring kernel can absorb (nets/architectures.py:529-546). */
int hno_act_fwd(const float *x, float *y, long long n, int act, void *stream);
int hno_act_bwd(const float *g, const float *y, float *gx, long long n, int act, void *stream);
/* y[b][c][v] = act(y[b][c][v] + bias[c]) in place: epilogue of 1x1x1 convolutions routed through hno_bmm (wide layers) */
int hno_bias_act(float *y, const float *bias, int B, int C, long long V, int act, void *stream);
int hno_add(const float *a, const float *b, float *out, long long n, void *stream);
/* dst[r][v] = v < V ? src[r][v] : 0 for v < ld_dst, r < rows: (b, c) volumes between the contiguous layout (ld = V) and the
 * channel-padded one (ld = V rounded up to 32 floats, padding zeroed) that the HNOSeg-XS path keeps its activations in */
int hno_chan_restride(const float *src, float *dst, long long rows, long long V, long long ld_src, long long ld_dst, void *stream);
/* fp32 <-> bf16 (round to nearest even) over n elements: the ends of a chain of blocks whose activations are bf16 in memory under
 * torch.autocast (HNO_ACT_IO16) -- the reference casts at the same places (autocast's to(bfloat16) in front of nn.Conv3d) */
/* TIMING PROBE, not a product path (csrc/hno_invpw.hip; LESSONS round 6): the memory traffic and instruction counts of a forward kernel
 * that computes the inverse plane transform inside the pointwise kernel that consumes it -- outputs are NOT the transform's. */
int hno_debug_invpw_fwd_probe(const float *Y, const float *t, const float *W, const float *bias, const float *TH, const float *TW,
                              float *u, float *xi, int B, long long V, float scale, int grid, void *stream);
int hno_cast_f32_bf16(const float *src, void *dst_bf16, long long n, void *stream);
int hno_cast_bf16_f32(const void *src_bf16, float *dst, long long n, void *stream);
/* out = alpha * a + beta * b (b may be NULL: out = alpha * a); the x +- x_reverse combinations of hartley_conv
 * (nets/hartley_operator.py:315-317) and gradient scaling in the data-parallel path */
int hno_axpby(float alpha, const float *a, float beta, const float *b, float *out, long long n, void *stream);

/* --------------------------------------------------------------- Pearson / Dice reductions
 * Labels are uint8 class indices (B, V) -- one-hot encoding (experiments/utils.py:74-97)
 * is fused.  stats is (B, K, 4) doubles: sum p, sum p^2, sum p*t, sum t.
 * kind 0: PCCLoss (nets/custom_losses.py:17-70), 1: DiceLoss (:73-111),
 * kind 2: ExpDiceLoss (:114-133, exponent `param`).
 * fwd: fills stats (workspace, B*K*4 doubles), coef (B,K,4) floats = {r or dice value, alpha,
 *      beta, gamma} with dloss/dprobs[b,k,v] = alpha*onehot + beta*p + gamma, and the scalar loss.
 * bwd: g_probs[b,k,v] = gscale[0] * dloss/dprobs from coef (gscale: device pointer to the
 *      upstream gradient, NULL = 1).
 */
int hno_loss_fwd(const float *probs, const uint8_t *labels, double *stats, float *coef, float *loss,
                 int B, int K, long long V, int kind, float param, void *stream);
/* the same with the statistics summed through per-workgroup rows instead of double atomics: no clear kernel, bit-reproducible loss.
 * workspace: hno_loss_workspace_doubles(B, K, V) doubles; its first B K 4 doubles hold the statistics afterwards.  Falls back to the
 * accumulating form when V % 4 != 0 or the pointers are not 16-byte aligned. */
size_t hno_loss_workspace_doubles(int B, int K, long long V);
int hno_loss_fwd_ws(const float *probs, const uint8_t *labels, double *workspace, size_t workspace_doubles, float *coef, float *loss,
                    int B, int K, long long V, int kind, float param, void *stream);
int hno_loss_bwd(const float *probs, const uint8_t *labels, const float *coef, const float *gscale,
                 float *g_probs, int B, int K, long long V, void *stream);

/* Round 4: the softmax head and the loss in one pass each way (nets/hnosegxs.py:174-180 + nets/custom_losses.py:17-133; in the
 * reference's loop `loss_fn(model(x), y)`, experiments/train_test.py:154-160).
 *   hno_uphead_loss_fwd: trilinear upsampling of the low-resolution logits + softmax -> probs, and the loss sums taken from the
 *     probabilities while they are in registers (no second pass over probs); coef / loss as hno_loss_fwd_ws leaves them.
 *     workspace: hno_uphead_loss_workspace_doubles(B, K) doubles.  Shapes: hno_uphead_loss_supported (even W <= 128, w <= 128, K <= 8).
 *   hno_upsoftmax_loss_bwd: the head's backward with d loss / d probs = gscale (alpha t + beta p + gamma) evaluated from coef and
 *     the labels in place of hno_loss_bwd's tensor; workspace of hno_upsoftmax_bwd_workspace_bytes (0 = not covered -> HNO_ELIMIT). */
int hno_uphead_loss_supported(int B, int K, int d, int h, int w, int D, int H, int W);
size_t hno_uphead_loss_workspace_doubles(int B, int K);
int hno_uphead_loss_fwd(const float *logits_lr, const unsigned char *labels, float *probs, double *workspace, size_t workspace_doubles,
                        float *coef, float *loss, int B, int K, int d, int h, int w, int D, int H, int W, long long ldlr, int kind,
                        float param, void *stream);
int hno_upsoftmax_loss_bwd(const float *probs, const unsigned char *labels, const float *coef, const float *gscale, float *g_lr,
                           void *workspace, int B, int K, int d, int h, int w, int D, int H, int W, long long ldlr, void *stream);

/* float labels (B,1,...) -> uint8 class indices with optional remap table (256 entries,
 * NULL = identity); one-hot output optional (NULL to skip). experiments/utils.py:74-119 */
int hno_labels_prepare(const float *labels_f32, const int *remap_from, const int *remap_to, int n_remap,
                       uint8_t *labels_u8, float *onehot, int B, int K, long long V, void *stream);

/* ---- deep-supervision convolution over T equally wide tensors (round 6) --------------------------------------------------------------
 * Replaces `torch.cat(tensors, dim=1)` + `conv_ds` (a k = 1 ConvNormAct, reference nets/architectures.py:341-343 and :196-199) and its
 * backward: out[b, k, v] = bias[k] + sum_t sum_c W[t][k][c] x_t[b, c, v] in ONE launch, its backward (all T input gradients, the
 * [T][K][C] weight gradient, the bias gradient) in one launch + one slab reduction.  x / gx: HOST arrays of T device pointers to
 * (B, C, ld) fp32 tensors (channel stride ld >= V, padding untouched; gx entries or the array may be NULL); W: [T][K][C] -- the
 * module's (K, T C) weight regrouped by leg.  hno_pwmulti_supported: the built shapes (K 2 ... 5, C 8 / 12 / 16 / 24, T <= 32). */
int hno_pwmulti_supported(int T, int C, int K);
int hno_pwmulti_fwd(const void *const *x, int T, int C, const float *W, const float *bias, float *out, int B, int K, long long V,
                    long long ld, void *stream);
size_t hno_pwmulti_bwd_workspace_bytes(int T, int C, int K, int B, long long V);
int hno_pwmulti_bwd(const float *g, const void *const *x, void *const *gx, int T, int C, const float *W, float *dW, float *dbias,
                    void *workspace, size_t workspace_bytes, int B, int K, long long V, long long ld, void *stream);

/* ------------------------------------------------------------------ optimizer
 * Multi-tensor Adamax: ONE launch updates every parameter (replaces torch.optim.Adamax.step as driven by
 * experiments/run.py:89-91 / train_test.py:171; arithmetic of torch/optim/adamax.py):
 *   g = grad * grad_scale + weight_decay * p;  m += (1 - beta1)(g - m);  u = max(beta2 u, |g| + eps);
 *   p -= lr / (1 - beta1^step) * m / u
 * `table` is a DEVICE array of n_chunks rows of hno_adamax_chunk_rows() (= 5) 64-bit words:
 *   { float *p, const float *grad, float *exp_avg, float *exp_inf, int64 n } -- a run of n elements of one tensor.
 * `step` is the 1-based step count.  grad_scale folds the 1/world of a SUM all-reduce into the update. */
int hno_adamax_chunk_rows(void);
int hno_adamax_multi(const void *table, int n_chunks, float lr, float beta1, float beta2, float eps,
                     float weight_decay, long long step, float grad_scale, void *stream);
/* Device-stepped form (round 4): step counter, learning-rate schedule and bias correction live in `state`, a DEVICE array of
 * hno_adamax_state_doubles() doubles (9 values, the kernel's ticket counter, the scheduler's tick count) { step, lr of the next update, base_lr, eta_min, T_cur, T_i, T_mult, schedule (0 constant,
 * 1 CosineAnnealingWarmRestarts stepped per optimizer step: experiments/run.py:92-103, train_test.py:173-174), clr (scratch) }.
 * A one-thread kernel advances the state, then the update runs with lr / (1 - beta1^step) read from it: no host value changes from
 * step to step, so the call can be captured into the training step's HIP graph (one graph replay per step and rank). */
int hno_adamax_state_doubles(void);
int hno_adamax_multi_dev(const void *table, int n_chunks, void *state, float beta1, float beta2, float eps, float weight_decay,
                         float grad_scale, void *stream);
/* The same under torch.amp.GradScaler (round 6; experiments/train_test.py:79,154-168 drive GradScaler.step(optimizer) / update()): the
 * optimizer declares `_step_supports_amp_scaling`, GradScaler then hands it two DEVICE scalars instead of synchronising on its inf
 * check -- `amp_scale` (the gradients are still multiplied by it: divided here, written back unscaled) and `found_inf` (non-zero: the
 * update is skipped, the step count stays, the schedule still ticks).  Either may be NULL.  state: hno_adamax_state_doubles() doubles,
 * [10] = scheduler ticks. */
int hno_adamax_multi_dev_amp(const void *table, int n_chunks, void *state, float beta1, float beta2, float eps, float weight_decay,
                             float grad_scale, const float *amp_scale, const float *found_inf, void *stream);
/* Join of two tensor sets (round 5): dst[i] = scale[i] * (a[i] + b[i]) elementwise for `count` fp32 tensors of n[i] elements; dst, a, b,
 * n, scale are HOST arrays (device pointers / sizes), dst[i] may alias a[i] or b[i].  One launch per 64 tensors, entries passed by value
 * in the kernel arguments: capturable into a HIP graph without a device table.  Replaces the torch._foreach_add_ / torch.lerp pair that
 * joined the two half-batch passes of a captured training step (experiments/train_test.py SampleSplit; no reference counterpart: the
 * reference runs the batch as one pass, train_test.py:154-170). */
int hno_sum_pairs(void *const *dst, const void *const *a, const void *const *b, const long long *n, const float *scale, int count,
                  void *stream);

/* ------------------------------------------------------------------ input pipeline
 * hno_zscore_modalities: x, out (C, V): per modality c, v = clip(x) if has_clip; statistics over v != mask_val if
 *   has_mask; out = (v - mean) / std (population std), masked voxels -> 0.  Replaces normalize_modalities /
 *   normalize_data (experiments/utils.py:25-71) as run.py:52-55 binds it (mask_val = 0).  mean_std (C, 2) receives the
 *   statistics (may be NULL).  A batch is C = B * modalities.  Workspace: hno_zscore_workspace_bytes(C).
 * hno_affine_nearest: x, out (C, D, H, W), out != x.  matrix12 is a HOST array, rows of [M | t] with
 *   (x, y, z)_in = M (x, y, z)_out + t in voxel indices: nearest-neighbour resampling (round half up), `cval`
 *   outside [-0.5, size - 0.5); then flips of the resampled image (bit 0 depth, bit 1 height, bit 2 width).
 *   Replaces apply_transform + flip_axis (experiments/data_io/dataset.py:205-244; SimpleITK AffineTransform +
 *   ResampleImageFilter with sitkNearestNeighbor).  2-D images: D = 1 and an identity z row. */
size_t hno_zscore_workspace_bytes(int C);
int hno_zscore_modalities(const float *x, float *out, float *mean_std, void *workspace, int C, long long V,
                          int has_mask, float mask_val, int has_clip, float clip_lo, float clip_hi, void *stream);
int hno_affine_nearest(const float *x, float *out, const double *matrix12, float cval, int flip_mask, int C, int D,
                       int H, int W, void *stream);

/* ------------------------------------------------------------------ deferred weight-gradient reduction
 * Every *_bwd entry point that produces a weight gradient writes per-workgroup partial slabs into its workspace and then
 * reduces them (fixed order, no atomics).  hno_set_defer_reduce(1) makes those calls RECORD the reduction instead (returns
 * the previous setting); hno_flush_reduces launches ONE kernel for everything recorded (bit-identical results).  Between
 * the call and the flush the caller must keep the workspace alive and must not read the gradient.  Per-call form: bit 8
 * (0x100) of `accumulate_gx` (hno_pwconv_bwd), `xa_act` (hno_pwconv_bwd_branch) or `residual` (hno_specmix_layers_bwd)
 * records just that call.  The Python layer uses the per-call bit during autograd's backward and flushes from the engine's
 * end-of-backward callback. */
int hno_set_defer_reduce(int on);
int hno_pending_reduces(void);
int hno_discard_reduces(void);   /* drop what was recorded (after an aborted backward pass); returns the count */
int hno_flush_reduces(void *stream);

/* ------------------------------------------------------------------ per-kernel profiler
 * hno_profile_begin arms HIP-event bracketing of every kernel launch (on the stream the kernel
 * is launched on); hno_profile_end stops it, waits for the events and returns the number of
 * records written as (kernel id, milliseconds, ALGORITHMIC bytes of that launch as defined in
 * DESIGN.md section 4).  Not for use during graph capture. */
int hno_profile_begin(int max_records);
int hno_profile_end(int *kernel_ids, float *ms, double *algorithmic_bytes, int capacity);
const char *hno_profile_kernel_name(int kernel_id);
/* ablation switches for kernel tuning (timing only: results are wrong when non-zero) */
int hno_set_debug(int flags);
/* with debug flag 64: clock64() stamps stored by thread 0 of workgroup 0 at the phase boundaries of the
 * instrumented kernels (tuning aid; n <= 64) */
int hno_debug_stamps(long long *out, int n);
/* test aid: the plane-kernel family the calling thread's last forward (inverse = 0) / inverse (1) transform launch took --
 * 0 none yet, 1 generic (a workgroup per plane), 2 the round-2 specialised kernels, 3 the LDS-DMA forward / half-plane item inverse
 * kernels of the 65 / 33 planes, 4 the item kernels for other plane sizes (hno_dht_items.hip, round 5).  The GPU tests pin the
 * families of the benchmark shapes and of the inference grid (a dropped instantiation falls back silently). */
int hno_debug_last_plane_family(int inverse);
/* test aid: weight-gradient slab reductions launched by this process so far -- batched = 0: one launch per slab set
 * (reduce_partials_kernel), 1: batched end-of-backward launches (reduce_partials_multi_kernel).  The GPU tests pin the launch count of a
 * captured training step with it (round 4's two-stream schedule silently fell back to 17 single launches per pass). */
long long hno_debug_reduce_launches(int batched);

/* ------------------------------------------------------------------------ self tests
 * C(MxN) = A(MxK) B(KxN) through the wave-level MFMA tile engine every kernel uses. */
int hno_selftest_gemm(const float *A, const float *Bm, float *C, int M, int N, int K, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HNO_H_ */
