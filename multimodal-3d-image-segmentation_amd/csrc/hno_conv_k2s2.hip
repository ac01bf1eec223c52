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
#include <stdlib.h>

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
    unsigned ldy;   // channel stride (floats) of y / gy / the saved output: Do * Ho * Wo, or padded to a multiple of 32 (ops.act_empty)
    // tiling of one (b, od) slab: Ho * wtr row tiles (32 consecutive ow of one output row) followed by
    // ncol * ctiles column tiles (32 consecutive oh at one of the last `ncol` columns).  Wo = 32 t + 1 (65, 33)
    // would otherwise spend a whole 32-wide tile per row on its single leftover column: 3 tiles per row, not 2.
    int wtr, ncol, ctiles, tiles_per_slab;
    // chained form (round 4): conv_in followed by conv1 = Conv3d(Cout -> C1, k = 1) + bias + act1 (nets/hnosegxs.py:151-152)
    const float *W1, *bias1;
    int C1, act1;
};

// tiles are 32 consecutive output voxels WITHIN one output row (ow), so the strided reads of a
// tile are one contiguous run; rows are enumerated as (b, od, oh).
// The patch rows arrive by LDS-DMA into a wave-private two-slot ring, one tile ahead of the products (hno_pwconv.hip's forward
// kernel, DESIGN.md lesson 27): a DMA instruction moves one k-step's operand (64 lanes x 4 bytes = taps kw = 0 / 1 of 32 output
// voxels, one contiguous 256-byte run in a row tile), the lane reads back its own 4 bytes.  A DMA cannot mask, so lanes outside
// the image (padding = 1) read a valid dummy element and are zeroed when the operand is read; ALWAYS KS_MAX instructions per
// tile, so that the counted wait is exact.  All tile arithmetic is 32-bit and wave-uniform where the tile allows it.
struct K2TileState {
    unsigned b, vo;       // batch index (uniform), output voxel offset of this lane
    bool live;            // lane has an output voxel
    bool ok[2][2];        // [kd][kh]: the tap row of this lane is inside the image
};

// CHAIN: the tile of conv_in's output never leaves the registers -- a 32x32x2 accumulator (column = voxel on the lane, 16 rows = channels
// crow32k(r, lane)) is the B operand of the next 32x32x2 product when conv1's weight columns are taken in that channel order
// (hno_pwconv.hip's chained kernels); y receives conv1's output and conv_in's is not written (the backward recomputes it from the image)
template <int KS_MAX, bool CHAIN = false, bool FIXED = false>  // >= Cin * 4 k-steps; FIXED: 4 -> 24 -> 24 channels, SELU twice (compile-time)
__global__ __launch_bounds__(256) void conv_k2s2_fwd_kernel(K2Args a_) {
    extern __shared__ float k2_ring[];                  // 4 waves x 2 slots x KS_MAX x 64 floats
    K2Args a = a_;
    if constexpr (FIXED) { a.Cin = 4; a.Cout = 24; a.C1 = 24; a.act = HNO_ACT_SELU; a.act1 = HNO_ACT_SELU; }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    {   // channel-padded output: the last workgroup zeroes the (< 32-float) padding behind every (b, c) volume -- nobody else writes it
        const unsigned Vo_ = (unsigned)a.Do * a.Ho * a.Wo;
        if (a.ldy > Vo_ && blockIdx.x == gridDim.x - 1) {
            const unsigned npad = a.ldy - Vo_, tot = (unsigned)(a.B * (CHAIN ? a.C1 : a.Cout)) * npad;
            for (unsigned i = threadIdx.x; i < tot; i += 256) a.y[(size_t)(i / npad) * a.ldy + Vo_ + i % npad] = 0.f;
        }
    }
    const int nks = a.Cin * 4;
    const int CY = CHAIN ? a.C1 : a.Cout;              // channels of the tensor this kernel writes
    float w2[CHAIN ? 16 : 1], bias2_r[CHAIN ? 16 : 1];
    if constexpr (CHAIN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = crow32k(r, lane);           // conv_in channel held in accumulator register r of this lane half
            w2[r] = (c < a.C1 && ch < a.Cout) ? a.W1[(size_t)c * a.Cout + ch] : 0.f;
            bias2_r[r] = (a.bias1 && ch < a.C1) ? a.bias1[ch] : 0.f;
        }
    }
    float w[KS_MAX];
#pragma unroll
    for (int ks = 0; ks < KS_MAX; ++ks) {
        const int k = 2 * ks + h;
        w[ks] = (ks < nks && c < a.Cout) ? a.W[(size_t)c * (a.Cin * 8) + k] : 0.f;
    }
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = crow32k(r, lane);
        bias_r[r] = (a.bias && o < a.Cout) ? a.bias[o] : 0.f;
    }
    const bool lin = a.act == HNO_ACT_NONE, exp_act = a.act == HNO_ACT_SELU || a.act == HNO_ACT_ELU;
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const unsigned ntiles = (unsigned)a.B * a.Do * a.tiles_per_slab;
    const unsigned Vo = (unsigned)a.Do * a.Ho * a.Wo, HW = (unsigned)a.H * a.Wd, DHW = (unsigned)a.D * HW;
    const unsigned nrow = (unsigned)a.Ho * a.wtr;
    float *ring = k2_ring + wave * (2 * KS_MAX * 64);
    const unsigned ring_b = (unsigned)(size_t)ring;
    const unsigned stride = gridDim.x * 4;
    auto issue = [&](unsigned t, int slot) {
        K2TileState ts;
        const unsigned slab = t / a.tiles_per_slab, idx = t - slab * a.tiles_per_slab;      // wave-uniform
        const unsigned b = slab / a.Do, od = slab - b * a.Do;
        int oh, ow;
        if (idx < nrow) {                  // row tile: 32 consecutive ow of output row oh
            const unsigned ohu = idx / a.wtr;
            oh = (int)ohu;
            ow = (int)(idx - ohu * a.wtr) * 32 + c;
            ts.live = ow < a.Wo - a.ncol;
        } else {                           // column tile: 32 consecutive oh at one of the last columns
            const unsigned j = idx - nrow, col = j / a.ctiles;
            oh = (int)(j - col * a.ctiles) * 32 + c;
            ow = a.Wo - a.ncol + (int)col;
            ts.live = oh < a.Ho;
        }
        ts.b = b;
        ts.vo = ts.live ? ((unsigned)od * a.Ho + oh) * a.Wo + ow : 0u;
        const int wl = 2 * ow - 1 + h;                               // input column of this lane (kw = lane half)
        const bool wok = ts.live && wl >= 0 && wl < a.Wd;
        unsigned rowoff[2];
        bool okh[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int hh = 2 * oh - 1 + kh;
            okh[kh] = wok && hh >= 0 && hh < a.H;
            rowoff[kh] = okh[kh] ? (unsigned)(hh * a.Wd + wl) * 4u : 0u;
        }
        const float *xb_ = a.x + (size_t)b * a.Cin * DHW;
        unsigned doff[2];                                           // element offset of depth plane d inside a channel (uniform)
        bool okd[2];
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) {
            const int d = 2 * (int)od - 1 + kd;
            okd[kd] = d >= 0 && d < a.D;                             // uniform
            doff[kd] = okd[kd] ? (unsigned)d * HW : 0u;
            ts.ok[kd][0] = okd[kd] && okh[0];
            ts.ok[kd][1] = okd[kd] && okh[1];
        }
#pragma unroll
        for (int ks = 0; ks < KS_MAX; ++ks) {
            const int kh = ks & 1, kd = (ks >> 1) & 1, i = ks >> 2;
            const bool oku = i < a.Cin && okd[kd];                   // uniform; otherwise a dummy row of channel 0
            const float *base = xb_ + (oku ? (unsigned)i * DHW + doff[kd] : 0u);      // < 2^29 elements (host check)
            dma_row_pair(base, rowoff[kh], __builtin_amdgcn_readfirstlane(ring_b + (slot * KS_MAX + ks) * 256));
        }
        return ts;
    };
    unsigned t = blockIdx.x * 4 + wave;
    K2TileState cur = {}, nxt = {};
    if (t < ntiles) cur = issue(t, 0);
    for (int slot = 0; t < ntiles; t += stride, slot ^= 1, cur = nxt) {
        static_assert(KS_MAX <= 63, "the DMA of one tile must fit the vmcnt counter");
        if (t + stride < ntiles) {
            nxt = issue(t + stride, slot ^ 1);
            dma_wait<KS_MAX>();            // only the next tile's DMA stays in flight
        } else {
            dma_wait<0>();
        }
        const float *sl = ring + slot * KS_MAX * 64 + lane;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS_MAX; ++ks)
            if (ks < nks) {
                const float xv = sl[ks * 64];
                acc = mfma32k(w[ks], cur.ok[(ks >> 1) & 1][ks & 1] ? xv : 0.f, acc);
            }
        float val[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) val[r] = acc[r] + bias_r[r];
        if (exp_act) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = neg_expm1(val[r]);
                asm volatile("" : "+v"(e));             // computed for every lane and selected: no branch per register
                val[r] = val[r] > 0.f ? ap * val[r] : aq * e;
            }
        } else if (!lin) {
#pragma unroll
            for (int r = 0; r < 16; ++r) val[r] = act_apply(val[r], a.act);
        }
        if constexpr (CHAIN) {
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) < a.Cout) acc2 = mfma32k(w2[r], val[r], acc2);      // (uniform: both halves' channels >= Cout)
            const bool exp2_act = a.act1 == HNO_ACT_SELU || a.act1 == HNO_ACT_ELU;
            const float ap2 = a.act1 == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
            const float aq2 = a.act1 == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) val[r] = acc2[r] + bias2_r[r];
            if (exp2_act) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float e = neg_expm1(val[r]);
                    asm volatile("" : "+v"(e));
                    val[r] = val[r] > 0.f ? ap2 * val[r] : aq2 * e;
                }
            } else if (a.act1 != HNO_ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) val[r] = act_apply(val[r], a.act1);
            }
        }
        float *y_l = a.y + (size_t)cur.b * CY * a.ldy + cur.vo + (h ? 4u * a.ldy : 0u);
        const bool full = __builtin_amdgcn_ballot_w64(cur.live) == ~0ull;
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o0 = (r & 3) + 8 * (r >> 2);
                if (o0 + 4 < CY) y_l[(size_t)o0 * a.ldy] = val[r];                 // uniform conditions
                else if (o0 < CY) { if (h == 0) y_l[(size_t)o0 * a.ldy] = val[r]; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o0 = (r & 3) + 8 * (r >> 2);
                if (cur.live && o0 + 4 * h < CY) y_l[(size_t)o0 * a.ldy] = val[r];
            }
        }
    }
}

// weight / bias gradient: dW[o][k] += sum_v g[o][v] * patch[k][v] with 16x16x4 MFMA from wave-private LDS tiles
// (K = 32 output voxels per tile).  The input image needs no gradient.
//   * the patch rows of the NEXT tile arrive by LDS-DMA (two-slot ring, row pairs K2_XP = 68 floats apart: conflict-free 16 x 4 operand
//     reads, see hno_pwconv.hip), gy / y of the next tile are register-prefetched behind the DMA; no workgroup barrier;
//   * padding taps are real zeros in this product (g is not zero there), so a lane whose tap lies outside the image overwrites its own
//     DMA'd element with 0 before the tile is read (every second row tile has one such lane: input column -1).
#define K2_LD 34
#define K2_XP 68
template <int KSO_MAX, int KCH>  // KSO_MAX >= ceil(Cout/2); KCH = 32-wide chunks of k = Cin*8
__global__ __launch_bounds__(256) void conv_k2s2_bwd_kernel(K2Args a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int nkso = (a.Cout + 1) / 2;
    const int K = a.Cin * 8;
    constexpr int KS = KCH * 16;                       // DMA instructions (k pairs) per tile
    constexpr int XS = KS * K2_XP;                     // floats per ring slot
    constexpr int WAVE_FLOATS = 32 * K2_LD + 2 * XS;
    float *G = lds + (size_t)wave * WAVE_FLOATS;       // [o][v]
    float *P = G + 32 * K2_LD;                         // two slots of [k / 2][k & 1][v]
    const unsigned p_lds = (unsigned)(size_t)P;
    for (int i = lane; i < WAVE_FLOATS; i += 64) G[i] = 0.f;
    float db[KSO_MAX];
#pragma unroll
    for (int ks = 0; ks < KSO_MAX; ++ks) db[ks] = 0.f;
    constexpr int MT = 2, NTK = KCH * 2;
    f32x4 dw[MT][NTK];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTK; ++n) dw[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned ntiles = (unsigned)a.B * a.Do * a.tiles_per_slab;
    const unsigned ngroups = (ntiles + 3) / 4;
    const unsigned Vo = (unsigned)a.Do * a.Ho * a.Wo, HW = (unsigned)a.H * a.Wd, DHW = (unsigned)a.D * HW;
    const unsigned nrow = (unsigned)a.Ho * a.wtr;
    const unsigned hoffV = h ? a.ldy : 0u;
    float pg[KSO_MAX], py[KSO_MAX];
    struct St { bool live; bool ok[2][2]; bool any_bad; };
    auto fetch = [&](unsigned grp, int slot) {
        St st;
        const unsigned t0 = grp * 4 + wave;
        const bool tlv = t0 < ntiles;                                            // uniform
        const unsigned t = tlv ? t0 : 0u;
        const unsigned slab = t / a.tiles_per_slab, idx = t - slab * a.tiles_per_slab;
        const unsigned b = slab / a.Do, od = slab - b * a.Do;
        int oh, ow;
        if (idx < nrow) {
            const unsigned ohu = idx / a.wtr;
            oh = (int)ohu;
            ow = (int)(idx - ohu * a.wtr) * 32 + c;
            st.live = tlv && ow < a.Wo - a.ncol;
        } else {
            const unsigned j = idx - nrow, col = j / a.ctiles;
            oh = (int)(j - col * a.ctiles) * 32 + c;
            ow = a.Wo - a.ncol + (int)col;
            st.live = tlv && oh < a.Ho;
        }
        const unsigned vo = st.live ? ((unsigned)od * a.Ho + oh) * a.Wo + ow : 0u;
        const int wl = 2 * ow - 1 + h;
        const bool wok = st.live && wl >= 0 && wl < a.Wd;
        unsigned rowoff[2];
        bool okh[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int hh = 2 * oh - 1 + kh;
            okh[kh] = wok && hh >= 0 && hh < a.H;
            rowoff[kh] = okh[kh] ? (unsigned)(hh * a.Wd + wl) * 4u : 0u;
        }
        bool bad = false;
        unsigned doff[2];
        bool okd[2];
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) {
            const int d = 2 * (int)od - 1 + kd;
            okd[kd] = d >= 0 && d < a.D;
            doff[kd] = okd[kd] ? (unsigned)d * HW : 0u;
            st.ok[kd][0] = okd[kd] && okh[0];
            st.ok[kd][1] = okd[kd] && okh[1];
            bad = bad || !st.ok[kd][0] || !st.ok[kd][1];
        }
        st.any_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;                   // uniform
        const float *xb_ = a.x + (size_t)b * a.Cin * DHW;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kd = (ks >> 1) & 1, i = ks >> 2;
            const bool oku = i < a.Cin && okd[kd];
            const float *base = xb_ + (oku ? (unsigned)i * DHW + doff[kd] : 0u);
            dma_row_pair(base, rowoff[ks & 1], __builtin_amdgcn_readfirstlane(p_lds + (slot * XS + ks * K2_XP) * 4));
        }
        // younger register loads: in-order completion makes hipcc's wait for them retire the DMA above as well
        const float *gy_b = a.gy + (size_t)b * a.Cout * a.ldy, *y_b = a.y_saved + (size_t)b * a.Cout * a.ldy;
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            pg[ks] = 0.f; py[ks] = 0.f;
            if (ks < nkso) {
                const int o0 = 2 * ks;
                const unsigned off = (o0 + 1 < a.Cout ? hoffV : 0u) + vo;
                pg[ks] = (gy_b + (size_t)o0 * a.ldy)[off];
                py[ks] = (y_b + (size_t)o0 * a.ldy)[off];
            }
        }
        return st;
    };
    St cur = {}, nxt = {};
    if (blockIdx.x < ngroups) cur = fetch(blockIdx.x, 0);
    int slot = 0;
    for (unsigned grp = blockIdx.x; grp < ngroups; grp += gridDim.x, slot ^= 1, cur = nxt) {
        float *Pc = P + slot * XS;
        dma_wait<0>();              // this tile's DMA and register loads are the only operations in flight (the kernel stores nothing)
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            if (ks < nkso) {
                float g = pg[ks] * act_grad_from_out(py[ks], a.act);
                g = (cur.live && 2 * ks + h < a.Cout) ? g : 0.f;
                G[(2 * ks + h) * K2_LD + c] = g;
                db[ks] += g;
            }
        }
        if (cur.any_bad) {          // zero padding taps (the DMA put a dummy element there)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (!cur.ok[(ks >> 1) & 1][ks & 1]) Pc[ks * K2_XP + lane] = 0.f;
        }
        if (grp + gridDim.x < ngroups) nxt = fetch(grp + gridDim.x, slot ^ 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            const float *ga = G + (lane & 15) * K2_LD + (lane >> 4);
            const float *pb = Pc + ((lane & 15) >> 1) * K2_XP + (lane & 1) * 32 + (lane >> 4);
#pragma unroll 2
            for (int ks = 0; ks < 8; ++ks) {
                float av[MT], bv[NTK];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = ga[m * 16 * K2_LD + ks * 4];
#pragma unroll
                for (int n = 0; n < NTK; ++n) bv[n] = pb[n * 8 * K2_XP + ks * 4];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTK; ++n) dw[m][n] = mfma16(av[m], bv[n], dw[m][n]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
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

// Backward of the chained stem (round 4): conv_in (k 2, s 2, p 1, Cin -> C0) + act, conv1 (k 1, C0 -> C1) + act1, one pass.
//   per tile of 32 output voxels:  g1 = g_y1 * act1'(y1)                       (registers; also an LDS tile [o][v])
//                                  g_x0 = W1^T g1                              (32x32x2, g1 registers are the B operand)
//                                  x0 = act(W_in patch + b_in)                  (RECOMPUTED from the DMA'd patch rows, same instruction
//                                                                               sequence as the forward: bit-identical; LDS tile [i][v])
//                                  dW1 += g1 x0^T, db1 += sum g1               (16x16x4 from the two LDS tiles)
//                                  p = g_x0 * act'(x0)                         (accumulator layout = x0's: elementwise in registers)
//                                  dW_in += p patch^T, db_in += sum p          (p overwrites the g1 tile; 16x16x4 as conv_k2s2_bwd_kernel)
// HBM: the image, g_y1 and y1 once (172 MB for 2 x 4 x 128^3 -> 24 x 65^3); apart, conv1's backward read g_y1, y1, x0 and wrote g_x0, and
// conv_in's read g_x0, x0 and the image (330 MB).  slab: [dW_in (C0 x K) | db_in (C0) | dW1 (C1 x C0) | db1 (C1)].
// FIXED: the HNOSeg-XS stem as every configuration of the reference builds it (4 -> 24 -> 24 channels, SELU twice) with all sizes and both
// activations as compile-time constants: the generic form spends a third of its instructions on per-k-step / per-activation branches
template <int KSO_MAX, bool FIXED>   // >= ceil(C1 / 2)
__global__ __launch_bounds__(256, 2) void conv_k2s2_chain_bwd_kernel(K2Args a_, int k2_xcd_off) {
    extern __shared__ float lds[];
    K2Args a = a_;
    if constexpr (FIXED) { a.Cin = 4; a.Cout = 24; a.C1 = 24; a.act = HNO_ACT_SELU; a.act1 = HNO_ACT_SELU; }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int C0 = a.Cout, C1 = a.C1;
    const int nks = a.Cin * 4, nks1 = (C1 + 1) / 2;
    const int K = a.Cin * 8;
    constexpr int KS = 16;                             // DMA instructions (k pairs) per tile: K <= 32
    constexpr int XS = KS * K2_XP;                     // floats per ring slot
    constexpr int WAVE_FLOATS = 2 * 32 * K2_LD + 2 * XS;
    float *G = lds + (size_t)wave * WAVE_FLOATS;       // [o][v]: g1, later p
    float *X0 = G + 32 * K2_LD;                        // [i][v]: recomputed conv_in output
    float *P = X0 + 32 * K2_LD;                        // two slots of [k / 2][k & 1][v]
    const unsigned p_lds = (unsigned)(size_t)P;
    for (int i = lane; i < WAVE_FLOATS; i += 64) G[i] = 0.f;
    float w[16], bias_r[16], w1t[KSO_MAX];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w[ks] = (ks < nks && c < C0) ? a.W[(size_t)c * K + 2 * ks + h] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = crow32k(r, lane);
        bias_r[r] = (a.bias && o < C0) ? a.bias[o] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < KSO_MAX; ++s) {
        const int o = 2 * s + h;
        w1t[s] = (s < nks1 && o < C1 && c < C0) ? a.W1[(size_t)o * C0 + c] : 0.f;
    }
    const bool exp_act = a.act == HNO_ACT_SELU || a.act == HNO_ACT_ELU;
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    float db1[KSO_MAX], dbin[16];
#pragma unroll
    for (int ks = 0; ks < KSO_MAX; ++ks) db1[ks] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dbin[r] = 0.f;
    f32x4 dwin[2][2], dw1[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) dwin[m][n] = dw1[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned ntiles = (unsigned)a.B * a.Do * a.tiles_per_slab;
    const unsigned ngroups = (ntiles + 3) / 4;
    const unsigned HW = (unsigned)a.H * a.Wd, DHW = (unsigned)a.D * HW;
    const unsigned nrow = (unsigned)a.Ho * a.wtr;
    const unsigned hoffV = h ? a.ldy : 0u;
    float pg[KSO_MAX], py[KSO_MAX];
    struct St { bool live; bool ok[2][2]; bool any_bad; };
    auto fetch = [&](unsigned grp, int slot) {
        St st;
        const unsigned t0 = grp * 4 + wave;
        const bool tlv = t0 < ntiles;                                            // uniform
        const unsigned t = tlv ? t0 : 0u;
        const unsigned slab = t / a.tiles_per_slab, idx = t - slab * a.tiles_per_slab;
        const unsigned b = slab / a.Do, od = slab - b * a.Do;
        int oh, ow;
        if (idx < nrow) {
            const unsigned ohu = idx / a.wtr;
            oh = (int)ohu;
            ow = (int)(idx - ohu * a.wtr) * 32 + c;
            st.live = tlv && ow < a.Wo - a.ncol;
        } else {
            const unsigned j = idx - nrow, col = j / a.ctiles;
            oh = (int)(j - col * a.ctiles) * 32 + c;
            ow = a.Wo - a.ncol + (int)col;
            st.live = tlv && oh < a.Ho;
        }
        const unsigned vo = st.live ? ((unsigned)od * a.Ho + oh) * a.Wo + ow : 0u;
        const int wl = 2 * ow - 1 + h;
        const bool wok = st.live && wl >= 0 && wl < a.Wd;
        unsigned rowoff[2];
        bool okh[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int hh = 2 * oh - 1 + kh;
            okh[kh] = wok && hh >= 0 && hh < a.H;
            rowoff[kh] = okh[kh] ? (unsigned)(hh * a.Wd + wl) * 4u : 0u;
        }
        bool bad = false;
        unsigned doff[2];
        bool okd[2];
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) {
            const int d = 2 * (int)od - 1 + kd;
            okd[kd] = d >= 0 && d < a.D;
            doff[kd] = okd[kd] ? (unsigned)d * HW : 0u;
            st.ok[kd][0] = okd[kd] && okh[0];
            st.ok[kd][1] = okd[kd] && okh[1];
            bad = bad || !st.ok[kd][0] || !st.ok[kd][1];
        }
        st.any_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;                   // uniform
        const float *xb_ = a.x + (size_t)b * a.Cin * DHW;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kd = (ks >> 1) & 1, i = ks >> 2;
            const bool oku = i < a.Cin && okd[kd];
            const float *base = xb_ + (oku ? (unsigned)i * DHW + doff[kd] : 0u);
            dma_row_pair(base, rowoff[ks & 1], __builtin_amdgcn_readfirstlane(p_lds + (slot * XS + ks * K2_XP) * 4));
        }
        // younger register loads: in-order completion makes hipcc's wait for them retire the DMA above as well
        const float *gy_b = a.gy + (size_t)b * C1 * a.ldy, *y_b = a.y_saved + (size_t)b * C1 * a.ldy;
        const unsigned offb = 4u * (hoffV + vo), offb_last = 4u * vo;       // (wave-uniform base + 32-bit lane offset: no 64-bit lane arithmetic)
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            pg[ks] = 0.f; py[ks] = 0.f;
            if (ks < nks1) {
                const int o0 = 2 * ks;
                const unsigned off = o0 + 1 < C1 ? offb : offb_last;
                pg[ks] = ld_off(gy_b + (size_t)o0 * a.ldy, off);
                py[ks] = ld_off(y_b + (size_t)o0 * a.ldy, off);
            }
        }
        return st;
    };
    St cur = {}, nxt = {};
    // XCD-aware order (round 6): a tile's image rows start one float before a 128-byte line, so every row of a tile shares a line with the
    // neighbouring tile (3 lines for 2: the 1.56x over-fetch of the round-4 PMC pass) -- and consecutive workgroups sit on different XCDs,
    // each with its own L2.  With the group count split into 8 contiguous ranges, one per XCD (workgroup i: XCD i % 8, its (i / 8)-th
    // worker), neighbouring tiles run on ONE XCD at about the same time and the shared line is an L2 hit.
    const bool xcd = (gridDim.x & 7) == 0 && ngroups >= 8u * gridDim.x && !k2_xcd_off;
    const unsigned R = (ngroups + 7) / 8, G8 = gridDim.x >> 3, xq = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const unsigned step = xcd ? G8 : gridDim.x, first = xcd ? jq : blockIdx.x;
    const unsigned lim = xcd ? (xq * R + R <= ngroups ? R : (xq * R < ngroups ? ngroups - xq * R : 0u)) : ngroups;
    const unsigned gbase = xcd ? xq * R : 0u;
    if (first < lim) cur = fetch(gbase + first, 0);
    int slot = 0;
    for (unsigned q = first; q < lim; q += step, slot ^= 1, cur = nxt) {
        float *Pc = P + slot * XS;
        dma_wait<0>();              // this tile's DMA and register loads are the only operations in flight (the kernel stores nothing)
        f32x16 accg, acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) accg[r] = acc0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            if (ks < nks1) {
                float g = pg[ks] * act_grad_from_out(py[ks], a.act1);
                g = (cur.live && 2 * ks + h < C1) ? g : 0.f;
                G[(2 * ks + h) * K2_LD + c] = g;
                db1[ks] += g;
                accg = mfma32k(w1t[ks], g, accg);       // g_x0 = W1^T g1
            }
        }
        if (cur.any_bad) {          // zero padding taps (the DMA put a dummy element there)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (!cur.ok[(ks >> 1) & 1][ks & 1]) Pc[ks * K2_XP + lane] = 0.f;
        }
        if (q + step < lim) nxt = fetch(gbase + q + step, slot ^ 1);
        // conv_in's output of this tile, as the forward computed it (a lane reads back only patch elements it may have zeroed itself)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            if (ks < nks) acc0 = mfma32k(w[ks], Pc[ks * K2_XP + lane], acc0);
        float x0v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x0v[r] = acc0[r] + bias_r[r];
        if (exp_act) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = neg_expm1(x0v[r]);
                asm volatile("" : "+v"(e));
                x0v[r] = x0v[r] > 0.f ? ap * x0v[r] : aq * e;
            }
        } else if (a.act != HNO_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) x0v[r] = act_apply(x0v[r], a.act);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) X0[crow32k(r, lane) * K2_LD + c] = x0v[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {   // dW1[o][i] += sum_v g1[o][v] x0[i][v]
            const float *ga = G + (lane & 15) * K2_LD + (lane >> 4);
            const float *xb = X0 + (lane & 15) * K2_LD + (lane >> 4);
#pragma unroll 2
            for (int ks = 0; ks < 8; ++ks) {
                float av[2], bv[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) av[m] = ga[m * 16 * K2_LD + ks * 4];
#pragma unroll
                for (int n = 0; n < 2; ++n) bv[n] = xb[n * 16 * K2_LD + ks * 4];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) dw1[m][n] = mfma16(av[m], bv[n], dw1[m][n]);
            }
        }
        // p = g_x0 * act'(x0): the gradient of conv_in's pre-activation, in x0's register layout
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pv[r] = accg[r] * act_grad_from_out(x0v[r], a.act);
            dbin[r] += pv[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) G[crow32k(r, lane) * K2_LD + c] = pv[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {   // dW_in[i][k] += sum_v p[i][v] patch[k][v]
            const float *ga = G + (lane & 15) * K2_LD + (lane >> 4);
            const float *pb = Pc + ((lane & 15) >> 1) * K2_XP + (lane & 1) * 32 + (lane >> 4);
#pragma unroll 2
            for (int ks = 0; ks < 8; ++ks) {
                float av[2], bv[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) av[m] = ga[m * 16 * K2_LD + ks * 4];
#pragma unroll
                for (int n = 0; n < 2; ++n) bv[n] = pb[n * 8 * K2_XP + ks * 4];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) dwin[m][n] = mfma16(av[m], bv[n], dwin[m][n]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    {
        const int n_in = C0 * K + C0, n = n_in + C1 * C0 + C1;
        __syncthreads();
        float *mine = lds + (size_t)wave * n;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = m * 16 + (lane >> 4) * 4 + r, k = nn * 16 + (lane & 15);
                    if (o < C0 && k < K) mine[o * K + k] = dwin[m][nn][r];
                    if (o < C1 && k < C0) mine[n_in + o * C0 + k] = dw1[m][nn][r];
                }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sv = dbin[r];
            for (int off = 16; off >= 1; off >>= 1) sv += __shfl_xor(sv, off);
            const int o = crow32k(r, lane);
            if (c == 0 && o < C0) mine[C0 * K + o] = sv;
        }
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            float sv = db1[ks];
            for (int off = 16; off >= 1; off >>= 1) sv += __shfl_xor(sv, off);
            const int o = 2 * ks + h;
            if (c == 0 && ks < nks1 && o < C1) mine[n_in + C1 * C0 + o] = sv;
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
                                 int D, int H, int Wd, int act, long long ldy, void *stream) {
    HNO_REQUIRE(x && W && y, "hno_conv_k2s2_fwd: null pointer");
    K2Args a = {};
    int rc = k2_fill(a, B, Cin, Cout, D, H, Wd, act);
    if (rc) return rc;
    const long long Vo_ = (long long)a.Do * a.Ho * a.Wo;
    HNO_REQUIRE(ldy == 0 || (ldy >= Vo_ && ldy < Vo_ + 64), "hno_conv_k2s2_fwd: channel stride %lld for %lld voxels", ldy, Vo_);
    a.ldy = (unsigned)(ldy ? ldy : Vo_);
    a.x = x; a.W = W; a.bias = bias; a.y = y;
    const long long ntiles = (long long)B * a.Do * a.tiles_per_slab;
    if (ntiles >= (1ll << 31) || (long long)Cin * D * H * Wd >= (1ll << 29) || (long long)Cout * a.Do * a.Ho * a.Wo >= (1ll << 29))
        return fail(HNO_ELIMIT, "hno_conv_k2s2_fwd: image of %d x %d x %d x %d exceeds the 32-bit offset range", Cin, D, H, Wd);
    long long grid = (ntiles + 3) / 4;
    if (grid > 1024) grid = 1024;   // four 4-wave workgroups per CU (measured: 256 -> 56 us, 512 -> 44, 1024 -> 40)
    if (debug_grid()) grid = debug_grid();
    {
        ProfScope _ps(KID_CONV_K2S2_FWD, (hipStream_t)stream, 4.0 * B * ((double)Cin * D * H * Wd + (double)Cout * a.Do * a.Ho * a.Wo));
        if (Cin <= 4) hipLaunchKernelGGL(conv_k2s2_fwd_kernel<16>, dim3((int)grid), dim3(256), 4 * 2 * 16 * 256, (hipStream_t)stream, a);
        else {
            static int attr = -1;
            if (attr != current_device()) { (void)hipFuncSetAttribute((const void *)conv_k2s2_fwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr = current_device(); }
            hipLaunchKernelGGL(conv_k2s2_fwd_kernel<32>, dim3((int)grid), dim3(256), 4 * 2 * 32 * 256, (hipStream_t)stream, a);
        }
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_conv_k2s2_bwd(const float *gy, const float *y, const float *x, const float *W, float *gx, float *dW,
                                 float *dbias, void *workspace, int B, int Cin, int Cout, int D, int H, int Wd, int act,
                                 long long ldy, void *stream) {
    HNO_REQUIRE(gy && y && x && dW && workspace, "hno_conv_k2s2_bwd: null pointer");
    (void)W;
    if (gx) return fail(HNO_ELIMIT, "hno_conv_k2s2_bwd: input-gradient (gx) is not implemented; the image input needs none");
    K2Args a = {};
    int rc = k2_fill(a, B, Cin, Cout, D, H, Wd, act);
    if (rc) return rc;
    {
        const long long Vo_ = (long long)a.Do * a.Ho * a.Wo;
        HNO_REQUIRE(ldy == 0 || (ldy >= Vo_ && ldy < Vo_ + 64), "hno_conv_k2s2_bwd: channel stride %lld for %lld voxels", ldy, Vo_);
        a.ldy = (unsigned)(ldy ? ldy : Vo_);
    }
    a.x = x; a.gy = gy; a.y_saved = y; a.dW = dW; a.dbias = dbias; a.partials = (float *)workspace;
    const long long ntiles = (long long)B * a.Do * a.tiles_per_slab;
    if (ntiles >= (1ll << 31) || (long long)Cin * D * H * Wd >= (1ll << 29) || (long long)Cout * a.Do * a.Ho * a.Wo >= (1ll << 29))
        return fail(HNO_ELIMIT, "hno_conv_k2s2_bwd: image of %d x %d x %d x %d exceeds the 32-bit offset range", Cin, D, H, Wd);
    long long grid = (ntiles + 3) / 4;
    if (grid > 512) grid = 512;     // two workgroups per CU (measured: 256 -> 79 us, 512 -> 61, 1024 -> 68 incl. the slab reduce)
    if (debug_grid()) grid = debug_grid();
    const int kch = Cin * 8 <= 32 ? 1 : 2;
    const size_t lds = sizeof(float) * 4 * (32 * K2_LD + 2 * kch * 16 * K2_XP);
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

static bool k2_fixed_off() {      // HNO_STEM_FIXED=0: the generic chained kernels also for the 4 -> 24 -> 24 / SELU stem (A/B)
    static const bool off = getenv("HNO_STEM_FIXED") && atoi(getenv("HNO_STEM_FIXED")) == 0;
    return off;
}

// ---- chained stem (round 4): conv_in + conv1 of HNOSeg-XS (nets/hnosegxs.py:102-108, 151-152) in one pass each way
extern "C" int hno_conv_k2s2_chain_supported(int Cin, int C0, int C1) {
    return Cin >= 1 && Cin <= 4 && C0 >= 1 && C0 <= 32 && C1 >= 1 && C1 <= 32;
}

#define K2_CHAIN_SLABS 512
extern "C" size_t hno_conv_k2s2_chain_bwd_workspace_bytes(int Cin, int C0, int C1) {
    if (!hno_conv_k2s2_chain_supported(Cin, C0, C1)) return 0;
    return sizeof(float) * K2_CHAIN_SLABS * ((size_t)C0 * Cin * 8 + C0 + (size_t)C1 * C0 + C1);
}

// y1 = act1(W1 act(conv_in(x) + b_in) + b1); conv_in's output is not written (hno_conv_k2s2_chain_bwd recomputes it)
extern "C" int hno_conv_k2s2_chain_fwd(const float *x, const float *W, const float *bias, const float *W1, const float *bias1, float *y1,
                                       int B, int Cin, int C0, int C1, int D, int H, int Wd, int act, int act1, long long ldy,
                                       void *stream) {
    HNO_REQUIRE(x && W && W1 && y1, "hno_conv_k2s2_chain_fwd: null pointer");
    if (!hno_conv_k2s2_chain_supported(Cin, C0, C1)) return fail(HNO_ELIMIT, "hno_conv_k2s2_chain_fwd: %d -> %d -> %d channels not covered", Cin, C0, C1);
    K2Args a = {};
    int rc = k2_fill(a, B, Cin, C0, D, H, Wd, act);
    if (rc) return rc;
    const long long Vo_ = (long long)a.Do * a.Ho * a.Wo;
    HNO_REQUIRE(ldy == 0 || (ldy >= Vo_ && ldy < Vo_ + 64), "hno_conv_k2s2_chain_fwd: channel stride %lld for %lld voxels", ldy, Vo_);
    a.ldy = (unsigned)(ldy ? ldy : Vo_);
    a.x = x; a.W = W; a.bias = bias; a.y = y1; a.W1 = W1; a.bias1 = bias1; a.C1 = C1; a.act1 = act1;
    const long long ntiles = (long long)B * a.Do * a.tiles_per_slab;
    if (ntiles >= (1ll << 31) || (long long)Cin * D * H * Wd >= (1ll << 29) || 32ll * a.Do * a.Ho * a.Wo >= (1ll << 29))
        return fail(HNO_ELIMIT, "hno_conv_k2s2_chain_fwd: image of %d x %d x %d x %d exceeds the 32-bit offset range", Cin, D, H, Wd);
    const bool fixed = Cin == 4 && C0 == 24 && C1 == 24 && act == HNO_ACT_SELU && act1 == HNO_ACT_SELU && !k2_fixed_off();
    long long grid = (ntiles + 3) / 4;
    const long long cap = fixed ? 1024 : 768;      // 120 / 164 registers: four / three 4-wave workgroups per CU are resident
    if (grid > cap) grid = cap;
    if (debug_grid()) grid = debug_grid();
    ProfScope _ps(KID_CONV_K2S2_FWD, (hipStream_t)stream, 4.0 * B * ((double)Cin * D * H * Wd + (double)C1 * a.Do * a.Ho * a.Wo));
    if (fixed)
        hipLaunchKernelGGL((conv_k2s2_fwd_kernel<16, true, true>), dim3((int)grid), dim3(256), 4 * 2 * 16 * 256, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv_k2s2_fwd_kernel<16, true, false>), dim3((int)grid), dim3(256), 4 * 2 * 16 * 256, (hipStream_t)stream, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// grads: [dW_in (C0 x Cin x 8) | db_in (C0) | dW1 (C1 x C0) | db1 (C1)]; bit 8 of act1: record the slab reduction for hno_flush_reduces
extern "C" int hno_conv_k2s2_chain_bwd(const float *gy1, const float *y1, const float *x, const float *W, const float *bias, const float *W1,
                                       float *grads, void *workspace, int B, int Cin, int C0, int C1, int D, int H, int Wd, int act,
                                       int act1, long long ldy, void *stream) {
    HNO_REQUIRE(gy1 && y1 && x && W && W1 && grads && workspace, "hno_conv_k2s2_chain_bwd: null pointer");
    const int defer_bit = (act1 >> 8) & 1;
    act1 &= 0xff;
    if (!hno_conv_k2s2_chain_supported(Cin, C0, C1)) return fail(HNO_ELIMIT, "hno_conv_k2s2_chain_bwd: %d -> %d -> %d channels not covered", Cin, C0, C1);
    K2Args a = {};
    int rc = k2_fill(a, B, Cin, C0, D, H, Wd, act);
    if (rc) return rc;
    const long long Vo_ = (long long)a.Do * a.Ho * a.Wo;
    HNO_REQUIRE(ldy == 0 || (ldy >= Vo_ && ldy < Vo_ + 64), "hno_conv_k2s2_chain_bwd: channel stride %lld for %lld voxels", ldy, Vo_);
    a.ldy = (unsigned)(ldy ? ldy : Vo_);
    a.x = x; a.W = W; a.bias = bias; a.W1 = W1; a.C1 = C1; a.act1 = act1;
    a.gy = gy1; a.y_saved = y1; a.partials = (float *)workspace;
    const long long ntiles = (long long)B * a.Do * a.tiles_per_slab;
    if (ntiles >= (1ll << 31) || (long long)Cin * D * H * Wd >= (1ll << 29) || 32ll * a.Do * a.Ho * a.Wo >= (1ll << 29))
        return fail(HNO_ELIMIT, "hno_conv_k2s2_chain_bwd: image of %d x %d x %d x %d exceeds the 32-bit offset range", Cin, D, H, Wd);
    long long grid = (ntiles + 3) / 4;
    if (grid > K2_CHAIN_SLABS) grid = K2_CHAIN_SLABS;     // two workgroups per CU
    if ((debug_grid()) && (debug_grid()) <= K2_CHAIN_SLABS) grid = debug_grid();
    const size_t lds = sizeof(float) * 4 * (2 * 32 * K2_LD + 2 * 16 * K2_XP);
    hipStream_t s = (hipStream_t)stream;
    static const int xcd_off = getenv("HNO_K2_XCD") ? !atoi(getenv("HNO_K2_XCD")) : 0;      // A/B aid: HNO_K2_XCD=0 -> the round-robin tile order
    {
        ProfScope _ps(KID_CONV_K2S2_BWD, s, 4.0 * B * ((double)Cin * D * H * Wd + 2.0 * C1 * a.Do * a.Ho * a.Wo));
        if (Cin == 4 && C0 == 24 && C1 == 24 && act == HNO_ACT_SELU && act1 == HNO_ACT_SELU && !k2_fixed_off()) {
            static int attr = -1;
            if (attr != current_device()) { (void)hipFuncSetAttribute((const void *)conv_k2s2_chain_bwd_kernel<12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = current_device(); }
            hipLaunchKernelGGL((conv_k2s2_chain_bwd_kernel<12, true>), dim3((int)grid), dim3(256), lds, s, a, xcd_off);
        } else if (C1 <= 24) {
            static int attr = -1;
            if (attr != current_device()) { (void)hipFuncSetAttribute((const void *)conv_k2s2_chain_bwd_kernel<12, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = current_device(); }
            hipLaunchKernelGGL((conv_k2s2_chain_bwd_kernel<12, false>), dim3((int)grid), dim3(256), lds, s, a, xcd_off);
        } else {
            static int attr = -1;
            if (attr != current_device()) { (void)hipFuncSetAttribute((const void *)conv_k2s2_chain_bwd_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = current_device(); }
            hipLaunchKernelGGL((conv_k2s2_chain_bwd_kernel<16, false>), dim3((int)grid), dim3(256), lds, s, a, xcd_off);
        }
    }
    HNO_CHECK_LAUNCH();
    const int n = C0 * Cin * 8 + C0 + C1 * C0 + C1;
    const int prev = hno_set_defer_reduce(0);
    struct Restore { int v; ~Restore() { hno_set_defer_reduce(v); } } restore{prev};
    hno_set_defer_reduce(defer_bit ? 1 : prev);
    return reduce_partials_launch(a.partials, (int)grid, n, grads, n, nullptr, s);
}
