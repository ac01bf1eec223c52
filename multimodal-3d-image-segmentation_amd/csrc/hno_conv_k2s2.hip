// Learnable 2x downsampling: Conv3d(Cin -> Cout, kernel 2, stride 2, padding 1) + bias + act.
//
// Reference: ConvNormAct(kernel_size=2, stride=2) as conv_in (nets/hnosegxs.py:102-104,151;
// nets/nets_utils.py:156-163: padding = kernel_size // 2 = 1), output size floor(N/2) + 1.
//
// Implicit GEMM on v_mfma_f32_32x32x2_f32: rows = output channels (weights in VGPRs),
// reduction index k = ((i*2 + kd)*2 + kh)*2 + kw, columns = 32 consecutive output voxels.
// Since kw is the lowest bit of k and the two lane halves of one MFMA operand register are
// k and k+1, one operand load reads x[2w-1] (half 0) and x[2w] (half 1) for 32 consecutive
// w: a single contiguous 256-byte run per instruction, no LDS, no wasted sector halves.
#include "hno_common.h"

namespace hno {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32k(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int crow32k(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

struct K2Args {
    const float *x, *W, *bias, *gy, *y_saved;
    float *y, *dW, *dbias, *partials;
    int B, Cin, Cout, D, H, Wd, Do, Ho, Wo;
    int act;
    // tiling of one (b, od) slab: Ho * wtr row tiles (32 consecutive ow of one output row) followed by
    // ncol * ctiles column tiles (32 consecutive oh at one of the last `ncol` columns).  Wo = 32 t + 1 (65, 33)
    // would otherwise spend a whole 32-wide tile per row on its single leftover column: 3 tiles per row, not 2.
    int wtr, ncol, ctiles, tiles_per_slab;
};

struct K2Tile {
    int b, od, oh, ow;   // oh / ow are per-lane
    bool live;
};
__device__ __forceinline__ K2Tile k2_tile(const K2Args &a, long long t, int c) {
    K2Tile r;
    const int idx = (int)(t % a.tiles_per_slab);
    const long long slab = t / a.tiles_per_slab;
    r.od = (int)(slab % a.Do);
    r.b = (int)(slab / a.Do);
    const int nrow = a.Ho * a.wtr;
    if (idx < nrow) {
        r.oh = idx / a.wtr;
        r.ow = (idx - r.oh * a.wtr) * 32 + c;
        r.live = r.ow < a.Wo - a.ncol;
    } else {
        const int j = idx - nrow, col = j / a.ctiles;
        r.oh = (j - col * a.ctiles) * 32 + c;
        r.ow = a.Wo - a.ncol + col;
        r.live = r.oh < a.Ho;
    }
    return r;
}

// x[b, i, 2od-1+kd, 2oh-1+kh, 2ow-1+kw] (zero outside).  k>>1 = (i,kd,kh) is wave-uniform, the lane
// half carries kw, so the address is  uniform row base + (2*ow - 1 + kw)  with a per-lane mask.
__device__ __forceinline__ float k2_patch(const K2Args &a, const float *xb_, int kpair, int od, int oh, int wl, bool wok) {
    const int kh = kpair & 1, kd = (kpair >> 1) & 1, i = kpair >> 2;
    const int d = 2 * od - 1 + kd, hh = 2 * oh - 1 + kh;       // oh (hence hh) is per-lane in column tiles
    if (!(i < a.Cin && d >= 0 && d < a.D)) return 0.f;          // uniform
    const bool ok = wok && hh >= 0 && hh < a.H;
    const float *plane = xb_ + ((size_t)i * a.D + d) * a.H * a.Wd;
    const float v = plane[ok ? hh * a.Wd + wl : 0];
    return ok ? v : 0.f;
}

// tiles are 32 consecutive output voxels WITHIN one output row (ow), so the strided reads of a
// tile are one contiguous run; rows are enumerated as (b, od, oh)
template <int KS_MAX>  // >= Cin * 4 k-steps
__global__ __launch_bounds__(256) void conv_k2s2_fwd_kernel(K2Args a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int nks = a.Cin * 4;
    float w[KS_MAX];
#pragma unroll
    for (int ks = 0; ks < KS_MAX; ++ks) {
        const int k = 2 * ks + h;
        w[ks] = (ks < nks && c < a.Cout) ? a.W[(size_t)c * (a.Cin * 8) + k] : 0.f;
    }
    const long long ntiles = (long long)a.B * a.Do * a.tiles_per_slab;
    const size_t Vo = (size_t)a.Do * a.Ho * a.Wo;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += (long long)gridDim.x * 4) {
        const K2Tile tl = k2_tile(a, t, c);
        const int b = tl.b, od = tl.od, oh = tl.oh, ow = tl.ow;
        const bool live = tl.live;
        const int wl = 2 * ow - 1 + h;                       // input column of this lane (kw = lane half)
        const bool wok = live && wl >= 0 && wl < a.Wd;
        const float *xb_ = a.x + (size_t)b * a.Cin * a.D * a.H * a.Wd;
        float xv[KS_MAX];
#pragma unroll
        for (int ks = 0; ks < KS_MAX; ++ks) xv[ks] = ks < nks ? k2_patch(a, xb_, ks, od, oh, wl, wok) : 0.f;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS_MAX; ++ks)
            if (ks < nks) acc = mfma32k(w[ks], xv[ks], acc);
        if (live) {
            const size_t vo = ((size_t)od * a.Ho + oh) * a.Wo + ow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = crow32k(r, lane);
                if (o < a.Cout) {
                    float val = acc[r] + (a.bias ? a.bias[o] : 0.f);
                    a.y[((size_t)b * a.Cout + o) * Vo + vo] = act_apply(val, a.act);
                }
            }
        }
    }
}

// weight / bias gradient: dW[o][k] += sum_v g[o][v] * patch[k][v] with 16x16x4 MFMA from a
// wave-private LDS tile (K = 32 output voxels per tile).  The input image needs no gradient.
#define K2_LD 34
template <int KSO_MAX, int KCH>  // KSO_MAX >= ceil(Cout/2); KCH = 32-wide chunks of k = Cin*8
__global__ __launch_bounds__(256) void conv_k2s2_bwd_kernel(K2Args a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int nkso = (a.Cout + 1) / 2;
    const int K = a.Cin * 8;
    float *G = lds + (size_t)wave * (32 + KCH * 32) * K2_LD;
    float *P = G + 32 * K2_LD;
    for (int i = lane; i < (32 + KCH * 32) * K2_LD; i += 64) G[i] = 0.f;
    float db[KSO_MAX];
#pragma unroll
    for (int ks = 0; ks < KSO_MAX; ++ks) db[ks] = 0.f;
    constexpr int MT = 2, NTK = KCH * 2;
    f32x4 dw[MT][NTK];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTK; ++n) dw[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long ntiles = (long long)a.B * a.Do * a.tiles_per_slab;
    const long long ngroups = (ntiles + 3) / 4;
    const size_t Vo = (size_t)a.Do * a.Ho * a.Wo;
    for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long long t = grp * 4 + wave;
        const bool tlv = t < ntiles;
        const K2Tile tl = k2_tile(a, tlv ? t : 0, c);
        const int b = tl.b, od = tl.od, oh = tl.oh, ow = tl.ow;
        const bool live = tlv && tl.live;
        const unsigned vo = live ? (unsigned)(((size_t)od * a.Ho + oh) * a.Wo + ow) : 0u;
        const int wl = 2 * ow - 1 + h;
        const bool wok = live && wl >= 0 && wl < a.Wd;
        const float *xb_ = a.x + (size_t)b * a.Cin * a.D * a.H * a.Wd;
        const float *gy_b = a.gy + (size_t)b * a.Cout * Vo, *y_b = a.y_saved + (size_t)b * a.Cout * Vo;
        const unsigned hoffV = h ? (unsigned)Vo : 0u;
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            float g = 0.f;
            if (ks < nkso) {
                const int o0 = 2 * ks;
                const unsigned off = (o0 + 1 < a.Cout ? hoffV : 0u) + vo;
                g = (gy_b + (size_t)o0 * Vo)[off] * act_grad_from_out((y_b + (size_t)o0 * Vo)[off], a.act);
                g = (live && o0 + h < a.Cout) ? g : 0.f;
                G[(o0 + h) * K2_LD + c] = g;
            }
            db[ks] += g;
        }
#pragma unroll 8
        for (int j = 0; j < KCH * 16; ++j)
            if (2 * j < K) P[(2 * j + h) * K2_LD + c] = k2_patch(a, xb_, j, od, oh, wl, wok);
        __syncthreads();
        {
            const float *ga = G + (lane & 15) * K2_LD + (lane >> 4);
            const float *pb = P + (lane & 15) * K2_LD + (lane >> 4);
#pragma unroll 2
            for (int ks = 0; ks < 8; ++ks) {
                float av[MT], bv[NTK];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = ga[m * 16 * K2_LD + ks * 4];
#pragma unroll
                for (int n = 0; n < NTK; ++n) bv[n] = pb[n * 16 * K2_LD + ks * 4];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTK; ++n) dw[m][n] = mfma16(av[m], bv[n], dw[m][n]);
            }
        }
        __syncthreads();
    }
    {
        const int n = a.Cout * K + a.Cout;
        __syncthreads();
        float *mine = lds + (size_t)wave * n;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < NTK; ++nn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = m * 16 + (lane >> 4) * 4 + r, k = nn * 16 + (lane & 15);
                    if (o < a.Cout && k < K) mine[o * K + k] = dw[m][nn][r];
                }
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            float sv = db[ks];
            for (int off = 16; off >= 1; off >>= 1) sv += __shfl_xor(sv, off);
            const int o = 2 * ks + h;
            if (c == 0 && ks < nkso && o < a.Cout) mine[a.Cout * K + o] = sv;
        }
        block_sum_to_slab(lds, n, a.partials + (size_t)blockIdx.x * n, threadIdx.x);
    }
}

}  // namespace hno

using namespace hno;

static int k2_fill(K2Args &a, int B, int Cin, int Cout, int D, int H, int Wd, int act) {
    HNO_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && Wd > 0, "hno_conv_k2s2: bad size");
    if (Cout > 32 || Cin > 8)
        return fail(HNO_ELIMIT, "hno_conv_k2s2: Cout=%d (max 32) / Cin=%d (max 8) outside the kernel limits", Cout, Cin);
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.Wd = Wd;
    a.Do = D / 2 + 1; a.Ho = H / 2 + 1; a.Wo = Wd / 2 + 1;
    a.act = act;
    const int rem = a.Wo % 32;
    a.ncol = (rem >= 1 && rem <= 4 && a.Wo > 32) ? rem : 0;      // few leftover columns: column tiles
    a.wtr = (a.Wo - a.ncol + 31) / 32;
    a.ctiles = (a.Ho + 31) / 32;
    a.tiles_per_slab = a.Ho * a.wtr + a.ncol * a.ctiles;
    return HNO_OK;
}

extern "C" int hno_conv_k2s2_fwd(const float *x, const float *W, const float *bias, float *y, int B, int Cin, int Cout,
                                 int D, int H, int Wd, int act, void *stream) {
    HNO_REQUIRE(x && W && y, "hno_conv_k2s2_fwd: null pointer");
    K2Args a = {};
    int rc = k2_fill(a, B, Cin, Cout, D, H, Wd, act);
    if (rc) return rc;
    a.x = x; a.W = W; a.bias = bias; a.y = y;
    const long long ntiles = (long long)B * a.Do * a.tiles_per_slab;
    long long grid = (ntiles + 3) / 4;
    if (grid > 1024) grid = 1024;   // 4 workgroups per CU; measured: 1024 -> 67 us, 2048 -> 71, 4096 -> 79, 512 -> 101
    if (debug_flags() >> 8) grid = debug_flags() >> 8;
    if (Cin <= 4)
        { ProfScope _ps(KID_CONV_K2S2_FWD, (hipStream_t)stream, 4.0 * B * ((double)Cin * D * H * Wd + (double)Cout * a.Do * a.Ho * a.Wo)); hipLaunchKernelGGL(conv_k2s2_fwd_kernel<16>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a); }
    else
        { ProfScope _ps(KID_CONV_K2S2_FWD, (hipStream_t)stream, 4.0 * B * ((double)Cin * D * H * Wd + (double)Cout * a.Do * a.Ho * a.Wo)); hipLaunchKernelGGL(conv_k2s2_fwd_kernel<32>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a); }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_conv_k2s2_bwd(const float *gy, const float *y, const float *x, const float *W, float *gx, float *dW,
                                 float *dbias, void *workspace, int B, int Cin, int Cout, int D, int H, int Wd, int act,
                                 void *stream) {
    HNO_REQUIRE(gy && y && x && dW && workspace, "hno_conv_k2s2_bwd: null pointer");
    (void)W;
    if (gx) return fail(HNO_ELIMIT, "hno_conv_k2s2_bwd: input-gradient (gx) is not implemented; the image input needs none");
    K2Args a = {};
    int rc = k2_fill(a, B, Cin, Cout, D, H, Wd, act);
    if (rc) return rc;
    a.x = x; a.gy = gy; a.y_saved = y; a.dW = dW; a.dbias = dbias; a.partials = (float *)workspace;
    const long long ntiles = (long long)B * a.Do * a.tiles_per_slab;
    long long grid = (ntiles + 3) / 4;
    if (grid > 1024) grid = 1024;
    if (debug_flags() >> 8) grid = debug_flags() >> 8;
    const int kch = Cin * 8 <= 32 ? 1 : 2;
    const size_t lds = sizeof(float) * 4 * (32 + kch * 32) * K2_LD;
    hipStream_t s = (hipStream_t)stream;
    if (Cout <= 24) {
        if (kch == 1) { ProfScope _ps(KID_CONV_K2S2_BWD, s, 4.0 * B * ((double)Cin * D * H * Wd + 2.0 * Cout * a.Do * a.Ho * a.Wo)); hipLaunchKernelGGL((conv_k2s2_bwd_kernel<12, 1>), dim3((int)grid), dim3(256), lds, s, a); }
        else { ProfScope _ps(KID_CONV_K2S2_BWD, s, 4.0 * B * ((double)Cin * D * H * Wd + 2.0 * Cout * a.Do * a.Ho * a.Wo)); hipLaunchKernelGGL((conv_k2s2_bwd_kernel<12, 2>), dim3((int)grid), dim3(256), lds, s, a); }
    } else {
        if (kch == 1) { ProfScope _ps(KID_CONV_K2S2_BWD, s, 4.0 * B * ((double)Cin * D * H * Wd + 2.0 * Cout * a.Do * a.Ho * a.Wo)); hipLaunchKernelGGL((conv_k2s2_bwd_kernel<16, 1>), dim3((int)grid), dim3(256), lds, s, a); }
        else { ProfScope _ps(KID_CONV_K2S2_BWD, s, 4.0 * B * ((double)Cin * D * H * Wd + 2.0 * Cout * a.Do * a.Ho * a.Wo)); hipLaunchKernelGGL((conv_k2s2_bwd_kernel<16, 2>), dim3((int)grid), dim3(256), lds, s, a); }
    }
    HNO_CHECK_LAUNCH();
    return reduce_partials_launch(a.partials, (int)grid, Cout * Cin * 8 + Cout, dW, Cout * Cin * 8, dbias, s);
}
