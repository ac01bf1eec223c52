// Output head (trilinear upsample + channel softmax), Pearson / Dice reductions and label
// preparation.  All are HBM-bound streaming kernels.
//
// Reference: F.interpolate(mode='trilinear') -> conv_out -> softmax (nets/hnosegxs.py:174-180;
// the 1x1x1 conv is applied at LOW resolution by hno_pwconv_fwd since it commutes with the
// per-channel linear interpolation), corrcoef/PCCLoss/dice_coef/DiceLoss/ExpDiceLoss
// (nets/custom_losses.py:17-133), to_categorical/remap_labels (experiments/utils.py:74-119).
#include <math.h>
#include <stdlib.h>

#include "hno_common.h"

namespace hno {

// PyTorch area_pixel_compute_source_index, align_corners=False, non-cubic:
//   src = max(scale * (dst + 0.5) - 0.5, 0), scale = in / out (fp32)
struct Lin {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ Lin lin_coord(int dst, float rscale, int in_size) {
    float src = rscale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lin l;
    l.i0 = (int)src;
    if (l.i0 > in_size - 1) l.i0 = in_size - 1;
    l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

struct UpArgs {
    const float *lr, *gp, *p;
    float *out;
    int B, K, d, h, w, D, H, W;
    float sd, sh, sw;
    int softmax;
    unsigned ldlr;     // floats between consecutive (b, k) volumes of the low-resolution tensor: d h w, or a channel-padded stride
};

template <int KMAX>
__global__ __launch_bounds__(256) void upsoftmax_fwd_kernel(UpArgs a) {
    const size_t V = (size_t)a.D * a.H * a.W, v_lr = (size_t)a.d * a.h * a.w;
    const size_t total = V * a.B;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = (int)(idx / V);
        const size_t v = idx % V;
        const int x = (int)(v % a.W), y = (int)((v / a.W) % a.H), z = (int)(v / ((size_t)a.W * a.H));
        const Lin lz = lin_coord(z, a.sd, a.d), ly = lin_coord(y, a.sh, a.h), lx = lin_coord(x, a.sw, a.w);
        const size_t o00 = ((size_t)lz.i0 * a.h + ly.i0) * a.w, o01 = ((size_t)lz.i0 * a.h + ly.i1) * a.w;
        const size_t o10 = ((size_t)lz.i1 * a.h + ly.i0) * a.w, o11 = ((size_t)lz.i1 * a.h + ly.i1) * a.w;
        float val[KMAX];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            val[k] = 0.f;
            if (k < a.K) {
                const float *s = a.lr + ((size_t)b * a.K + k) * a.ldlr;
                // same association order as upsample_trilinear3d: w-lerp, then h, then d
                const float c00 = lx.w0 * s[o00 + lx.i0] + lx.w1 * s[o00 + lx.i1];
                const float c01 = lx.w0 * s[o01 + lx.i0] + lx.w1 * s[o01 + lx.i1];
                const float c10 = lx.w0 * s[o10 + lx.i0] + lx.w1 * s[o10 + lx.i1];
                const float c11 = lx.w0 * s[o11 + lx.i0] + lx.w1 * s[o11 + lx.i1];
                val[k] = lz.w0 * (ly.w0 * c00 + ly.w1 * c01) + lz.w1 * (ly.w0 * c10 + ly.w1 * c11);
                mx = fmaxf(mx, val[k]);
            }
        }
        if (a.softmax) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < a.K) {
                    val[k] = expf(val[k] - mx);
                    sum += val[k];
                }
            const float inv = 1.f / sum;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) val[k] *= inv;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < a.K) a.out[((size_t)b * a.K + k) * V + v] = val[k];
    }
}

// Wave-per-row-segment form of the two head kernels (probabilities / arg max): a wave owns 64 consecutive x of one output row, so
// (b, z, y), the row offsets and the y / z interpolation weights are wave-uniform -- the one-thread-per-voxel form above spends most of
// its instructions on five 64-bit divisions per voxel.  Same loads (consecutive lanes -> consecutive x: one or two cache lines per
// gather), same arithmetic and association order.
template <int KMAX, bool ARGMAX>
__global__ __launch_bounds__(256) void uphead_seg_kernel(UpArgs a, unsigned char *__restrict__ labels) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nseg = (unsigned)(a.W + 63) / 64, items = (unsigned)a.B * a.D * a.H * nseg;
    const unsigned v_lr = (unsigned)a.d * a.h * a.w;
    const size_t V = (size_t)a.D * a.H * a.W;
    for (unsigned item = blockIdx.x * 4 + wave; item < items; item += gridDim.x * 4) {
        const unsigned row = item / nseg, seg = item - row * nseg;             // wave-uniform
        const unsigned bz = row / a.H, y = row - bz * a.H, b = bz / a.D, z = bz - b * a.D;
        const int x = (int)seg * 64 + lane;
        const bool in = x < a.W;
        const Lin lz = lin_coord((int)z, a.sd, a.d), ly = lin_coord((int)y, a.sh, a.h), lx = lin_coord(in ? x : a.W - 1, a.sw, a.w);
        const unsigned o00 = ((unsigned)lz.i0 * a.h + ly.i0) * a.w, o01 = ((unsigned)lz.i0 * a.h + ly.i1) * a.w;
        const unsigned o10 = ((unsigned)lz.i1 * a.h + ly.i0) * a.w, o11 = ((unsigned)lz.i1 * a.h + ly.i1) * a.w;
        float val[KMAX];
        float mx = -3.0e38f;
        int arg = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            val[k] = 0.f;
            if (k < a.K) {
                const float *s = a.lr + ((size_t)b * a.K + k) * a.ldlr;
                // same association order as upsample_trilinear3d: w-lerp, then h, then d
                const float c00 = lx.w0 * s[o00 + lx.i0] + lx.w1 * s[o00 + lx.i1];
                const float c01 = lx.w0 * s[o01 + lx.i0] + lx.w1 * s[o01 + lx.i1];
                const float c10 = lx.w0 * s[o10 + lx.i0] + lx.w1 * s[o10 + lx.i1];
                const float c11 = lx.w0 * s[o11 + lx.i0] + lx.w1 * s[o11 + lx.i1];
                val[k] = lz.w0 * (ly.w0 * c00 + ly.w1 * c01) + lz.w1 * (ly.w0 * c10 + ly.w1 * c11);
                if (ARGMAX) {
                    if (val[k] > mx) { mx = val[k]; arg = k; }
                } else {
                    mx = fmaxf(mx, val[k]);
                }
            }
        }
        const size_t vo = ((size_t)z * a.H + y) * a.W + x;
        if constexpr (ARGMAX) {
            if (in) labels[(size_t)b * V + vo] = (unsigned char)arg;
        } else {
            if (a.softmax) {
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < a.K) {
                        val[k] = expf(val[k] - mx);
                        sum += val[k];
                    }
                const float inv = 1.f / sum;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) val[k] *= inv;
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < a.K && in) a.out[((size_t)b * a.K + k) * V + vo] = val[k];
        }
    }
}

// Row form of the head (round 4): a wave owns whole output rows (W <= 128: lane l holds x = 2 l, 2 l + 1) and interpolates in SOURCE
// space first -- the y / z blend of the four source rows is done once per source column (lanes = source columns, 4 coalesced loads +
// 3 fma per class) and parked in a wave-private LDS row; every output voxel then takes its two x taps from LDS.  The voxel form above
// gathers 8 values and blends 7 times per (voxel, class): ~200 VALU instructions per voxel, which is what bounded it (36 us for 76 MB);
// this form needs ~45.  (Blending y / z before x instead of x first changes the rounding, not the result: <= 2 ulp.)
// STATS: the four loss sums per (sample, class) -- sum p, sum p^2, sum p t, sum t (nets/custom_losses.py:17-41, 73-90) -- are taken from
// the probabilities while they are in registers (the separate statistics pass re-read all 67 MB): fp32 per lane over the wave's <= 16
// rows, then fp64 across lanes, waves and workgroups in a fixed order; row [b][workgroup][K * 4] of `rows` goes to loss_finalize_kernel.
#define UPR_WAVES 8
#define UPR_PITCH 132          // w + 1 <= 129 source columns (+ the duplicated last one) per class

template <int KMAX, bool STATS, bool EXACT>      // EXACT: K == KMAX at compile time (no per-class branches)
__global__ __launch_bounds__(64 * UPR_WAVES) void uphead_rows_kernel(UpArgs a, const uint8_t *__restrict__ lab, double *__restrict__ rows) {
    constexpr int NT = 64 * UPR_WAVES, NS = KMAX * 4;
    // the y / z-blended source rows of the waves; the statistics' cross-lane stage reuses the space afterwards
    constexpr int NV = UPR_WAVES * KMAX * UPR_PITCH, NR = STATS ? NS * NT : 1;
    __shared__ float lds[NV > NR ? NV : NR];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y, K = EXACT ? KMAX : a.K, W = a.W, w = a.w;
    const size_t V = (size_t)a.D * a.H * W;
    const int nrow = a.D * a.H;
    float *Vw = lds + wave * (KMAX * UPR_PITCH);
    // x taps of this lane's two voxels (the same for every row)
    const int x0 = 2 * lane;
    const bool in = x0 < W;                                   // W is even: both voxels or none
    const Lin le = lin_coord(in ? x0 : 0, a.sw, w), lo = lin_coord(in ? x0 + 1 : 0, a.sw, w);
    // source columns of this lane: lane and lane + 64, clamped to w - 1.  Lanes beyond the row thus hold the LAST column, and writing
    // every lane's value unconditionally leaves Vw[w] = Vw[w - 1]: the tap i1 = i0 + 1 is always readable (torch clamps i1 to w - 1)
    const bool two = w >= 64;
    const unsigned j1b = 4u * (unsigned)(lane < w ? lane : w - 1), j2b = 4u * (unsigned)(lane + 64 < w ? lane + 64 : w - 1);
    const unsigned xb = 8u * (unsigned)lane;
    float sp[KMAX], sp2[KMAX], spt[KMAX];
    int st[KMAX];                                             // label counts of the WAVE (scalar: popcount of the compare masks)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { sp[k] = sp2[k] = spt[k] = 0.f; st[k] = 0; }
    // rows of this workgroup and of this wave: even split at row granularity (the grid is sized to the residency of the chip)
    // XCD-aware row ranges (round 5): workgroups go to the 8 XCDs round-robin, and an output row reads the four source rows its
    // neighbours in y AND in z read too.  With consecutive ranges on consecutive workgroups every XCD touched every source slice and
    // fetched it into its own L2: 128.6 MB of traffic for 80 MB of algorithmic bytes (profiles/hbm_traffic.json, round 4).  Workgroup
    // i now takes range (i % 8) * (G / 8) + i / 8: one XCD owns a contiguous eighth of the rows of a sample.
    const unsigned G = gridDim.x, bx = (G & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3);
    const int wg0 = (int)((long long)nrow * bx / G), wg1 = (int)((long long)nrow * (bx + 1) / G);
    const int r0 = wg0 + (wg1 - wg0) * wave / UPR_WAVES, r1 = wg0 + (wg1 - wg0) * (wave + 1) / UPR_WAVES;
    float raw1[KMAX][4], raw2[KMAX][4];
    unsigned l2n = 0;
    Lin lz, ly;
    auto request = [&](int r) {                               // the four source rows of output row r (wave-uniform offsets)
        const int z = r / a.H, y = r - z * a.H;
        lz = lin_coord(z, a.sd, a.d);
        ly = lin_coord(y, a.sh, a.h);
        const unsigned o00 = 4u * (((unsigned)lz.i0 * a.h + ly.i0) * w), o01 = 4u * (((unsigned)lz.i0 * a.h + ly.i1) * w);
        const unsigned o10 = 4u * (((unsigned)lz.i1 * a.h + ly.i0) * w), o11 = 4u * (((unsigned)lz.i1 * a.h + ly.i1) * w);
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const float *s = a.lr + ((size_t)b * K + k) * a.ldlr;
                raw1[k][0] = ld_off(s, o00 + j1b); raw1[k][1] = ld_off(s, o01 + j1b);
                raw1[k][2] = ld_off(s, o10 + j1b); raw1[k][3] = ld_off(s, o11 + j1b);
                if (two) {
                    raw2[k][0] = ld_off(s, o00 + j2b); raw2[k][1] = ld_off(s, o01 + j2b);
                    raw2[k][2] = ld_off(s, o10 + j2b); raw2[k][3] = ld_off(s, o11 + j2b);
                }
            }
        if (STATS && in)
            l2n = *reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(lab + (size_t)b * V + (size_t)r * W) + 2u * (unsigned)lane);
    };
    if (r0 < r1) request(r0);
    for (int r = r0; r < r1; ++r) {
        // ---- y / z blend per source column -> LDS
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                float *vs = Vw + k * UPR_PITCH;
                vs[lane] = lz.w0 * (ly.w0 * raw1[k][0] + ly.w1 * raw1[k][1]) + lz.w1 * (ly.w0 * raw1[k][2] + ly.w1 * raw1[k][3]);
                if (two) vs[lane + 64] = lz.w0 * (ly.w0 * raw2[k][0] + ly.w1 * raw2[k][1]) + lz.w1 * (ly.w0 * raw2[k][2] + ly.w1 * raw2[k][3]);
            }
        const unsigned l2 = l2n;
        if (r + 1 < r1) request(r + 1);                       // in flight under the taps, the softmax and the stores of this row
        // ---- x taps, softmax, store
        float pe[KMAX], po[KMAX];
        float me = -3.0e38f, mo = -3.0e38f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            pe[k] = po[k] = 0.f;
            if (k < K) {
                const float *vs = Vw + k * UPR_PITCH;
                pe[k] = le.w0 * vs[le.i0] + le.w1 * vs[le.i0 + 1];
                po[k] = lo.w0 * vs[lo.i0] + lo.w1 * vs[lo.i0 + 1];
                me = fmaxf(me, pe[k]);
                mo = fmaxf(mo, po[k]);
            }
        }
        if (a.softmax) {
            // e^x = exp2(x log2 e) on the hardware exp2: the rounded product costs |x| 2^-24 relative, i.e. < 3e-8 absolute on e^x for x <= 0
            float se = 0.f, so = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    pe[k] = __builtin_amdgcn_exp2f((pe[k] - me) * 1.44269504f);
                    po[k] = __builtin_amdgcn_exp2f((po[k] - mo) * 1.44269504f);
                    se += pe[k];
                    so += po[k];
                }
            float ie = __builtin_amdgcn_rcpf(se), io = __builtin_amdgcn_rcpf(so);
            ie = fmaf(fmaf(-se, ie, 1.f), ie, ie);
            io = fmaf(fmaf(-so, io, 1.f), io, io);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                pe[k] *= ie;
                po[k] *= io;
            }
        }
        if (in) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    float *orow = a.out + ((size_t)b * K + k) * V + (size_t)r * W;          // wave-uniform
                    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(orow) + xb) = make_float2(pe[k], po[k]);
                }
        }
        if constexpr (STATS) {
            const int l0 = in ? (int)(l2 & 255) : -1, l1 = in ? (int)(l2 >> 8) : -1;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    const bool te = l0 == k, to = l1 == k;
                    sp[k] += pe[k] + po[k];
                    sp2[k] = fmaf(pe[k], pe[k], fmaf(po[k], po[k], sp2[k]));
                    spt[k] += (te ? pe[k] : 0.f) + (to ? po[k] : 0.f);
                    st[k] += __builtin_popcountll(__ballot(te)) + __builtin_popcountll(__ballot(to));
                }
        }
    }
    if constexpr (STATS) {
        __syncthreads();                                      // every wave is done with its rows of `lds`
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            lds[(k * 4 + 0) * NT + threadIdx.x] = in ? sp[k] : 0.f;
            lds[(k * 4 + 1) * NT + threadIdx.x] = in ? sp2[k] : 0.f;
            lds[(k * 4 + 2) * NT + threadIdx.x] = spt[k];
            lds[(k * 4 + 3) * NT + threadIdx.x] = lane == 0 ? (float)st[k] : 0.f;       // <= 2 * 64 * rows of the wave: exact
        }
        __syncthreads();
        constexpr int NSP = KMAX <= 2 ? 8 : (KMAX <= 4 ? 16 : 32), TPS = NT / NSP;     // threads per statistic (a power of two <= 64)
        const int sidx = threadIdx.x / TPS, t = threadIdx.x - sidx * TPS;
        double acc = 0.0;
        if (sidx < NS) {
#pragma unroll 4
            for (int i = 0; i < NSP; ++i) acc += (double)lds[sidx * NT + t + i * TPS];
        }
        for (int off = TPS / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
        if (t == 0 && sidx < K * 4) rows[((size_t)b * gridDim.x + blockIdx.x) * (K * 4) + sidx] = acc;
    }
}

// Inference head: argmax over channels of the upsampled logits, written as uint8 class labels.  softmax is
// monotone, so the probabilities (67 MB per 2 volumes, and their trip over PCIe in the reference's testing loop,
// experiments/train_test.py:398-408) are never formed.  Ties resolve to the lowest index like numpy.argmax.
template <int KMAX>
__global__ __launch_bounds__(256) void upargmax_kernel(UpArgs a, unsigned char *__restrict__ labels) {
    const size_t V = (size_t)a.D * a.H * a.W, v_lr = (size_t)a.d * a.h * a.w;
    const size_t total = V * a.B;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = (int)(idx / V);
        const size_t v = idx % V;
        const int x = (int)(v % a.W), y = (int)((v / a.W) % a.H), z = (int)(v / ((size_t)a.W * a.H));
        const Lin lz = lin_coord(z, a.sd, a.d), ly = lin_coord(y, a.sh, a.h), lx = lin_coord(x, a.sw, a.w);
        const size_t o00 = ((size_t)lz.i0 * a.h + ly.i0) * a.w, o01 = ((size_t)lz.i0 * a.h + ly.i1) * a.w;
        const size_t o10 = ((size_t)lz.i1 * a.h + ly.i0) * a.w, o11 = ((size_t)lz.i1 * a.h + ly.i1) * a.w;
        float best = -3.0e38f;
        int arg = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < a.K) {
                const float *s = a.lr + ((size_t)b * a.K + k) * a.ldlr;
                const float c00 = lx.w0 * s[o00 + lx.i0] + lx.w1 * s[o00 + lx.i1];
                const float c01 = lx.w0 * s[o01 + lx.i0] + lx.w1 * s[o01 + lx.i1];
                const float c10 = lx.w0 * s[o10 + lx.i0] + lx.w1 * s[o10 + lx.i1];
                const float c11 = lx.w0 * s[o11 + lx.i0] + lx.w1 * s[o11 + lx.i1];
                const float val = lz.w0 * (ly.w0 * c00 + ly.w1 * c01) + lz.w1 * (ly.w0 * c10 + ly.w1 * c11);
                if (val > best) {
                    best = val;
                    arg = k;
                }
            }
        }
        labels[idx] = (unsigned char)arg;
    }
}

// first / last high-res index whose i0 is >= lo_i / <= hi_i (i0 is monotone in dst)
__device__ __forceinline__ void contrib_range(int i, float rscale, int in_size, int out_size, int &first, int &last) {
    // dst contributes to low-res index i  <=>  i0(dst) in {i-1, i}
    const float inv = 1.f / rscale;
    int f = (int)floorf(((float)(i - 1) + 0.5f) * inv - 0.5f);
    f = f < 0 ? 0 : (f > out_size - 1 ? out_size - 1 : f);
    while (f > 0 && lin_coord(f - 1, rscale, in_size).i0 >= i - 1) --f;
    while (f < out_size - 1 && lin_coord(f, rscale, in_size).i0 < i - 1) ++f;
    int l = (int)ceilf(((float)(i + 1) + 0.5f) * inv - 0.5f);
    l = l < 0 ? 0 : (l > out_size - 1 ? out_size - 1 : l);
    while (l < out_size - 1 && lin_coord(l + 1, rscale, in_size).i0 <= i) ++l;
    while (l > 0 && lin_coord(l, rscale, in_size).i0 > i) --l;
    first = f;
    last = l;
}
__device__ __forceinline__ float lin_weight(const Lin &l, int i) { return (l.i0 == i ? l.w0 : 0.f) + (l.i1 == i ? l.w1 : 0.f); }

// g_lr[b,k,i] = sum over high-res voxels of  trilinear weight * softmax'(p, gp)_k   (gather form)
template <int KMAX>
__global__ __launch_bounds__(256) void upsoftmax_bwd_kernel(UpArgs a) {
    const size_t V = (size_t)a.D * a.H * a.W, v_lr = (size_t)a.d * a.h * a.w;
    const size_t total = v_lr * a.B;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = (int)(idx / v_lr);
        const size_t v = idx % v_lr;
        const int ix = (int)(v % a.w), iy = (int)((v / a.w) % a.h), iz = (int)(v / ((size_t)a.w * a.h));
        int z0, z1, y0, y1, x0, x1;
        contrib_range(iz, a.sd, a.d, a.D, z0, z1);
        contrib_range(iy, a.sh, a.h, a.H, y0, y1);
        contrib_range(ix, a.sw, a.w, a.W, x0, x1);
        float acc[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
        for (int z = z0; z <= z1; ++z) {
            const float wz = lin_weight(lin_coord(z, a.sd, a.d), iz);
            if (wz == 0.f) continue;
            for (int y = y0; y <= y1; ++y) {
                const float wy = lin_weight(lin_coord(y, a.sh, a.h), iy) * wz;
                if (wy == 0.f) continue;
                for (int x = x0; x <= x1; ++x) {
                    const float wgt = lin_weight(lin_coord(x, a.sw, a.w), ix) * wy;
                    if (wgt == 0.f) continue;
                    const size_t hv = ((size_t)z * a.H + y) * a.W + x;
                    float g[KMAX], pr[KMAX];
                    float dot = 0.f;
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        g[k] = 0.f;
                        pr[k] = 0.f;
                        if (k < a.K) {
                            g[k] = a.gp[((size_t)b * a.K + k) * V + hv];
                            if (a.softmax) {
                                pr[k] = a.p[((size_t)b * a.K + k) * V + hv];
                                dot += pr[k] * g[k];
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) acc[k] += wgt * (a.softmax ? pr[k] * (g[k] - dot) : g[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < a.K) a.out[((size_t)b * a.K + k) * a.ldlr + v] = acc[k];
    }
}

// ---- separable backward --------------------------------------------------------------------
// trilinear^T = (D-axis)^T (H-axis)^T (W-axis)^T.  The gather kernel above re-reads every high-res
// voxel ~8 times through L2 (measured 749 MB of HBM/L2 traffic for 143 MB of operands).  Here the
// 134 MB of (g, p) are streamed exactly once:
//   plane kernel: one workgroup per (b, z_hr, band of low-res rows).  Batches of UPB_ROWS high-res
//     rows are loaded with 16-byte loads (prefetched one batch ahead), the softmax gradient is formed
//     per voxel and staged in LDS, then thread (k, j) folds the W axis (<= UPB_MAXT taps held in
//     registers) and the H axis (rows arrive in order, so a 2-row sliding window in registers is
//     enough: no LDS accumulator, no atomics) and emits T[b][k][z_hr][i][j];
//   D kernel: g_lr[b][k][iz][i][j] = sum of <= ~4 planes of T (17 MB, L2/MALL resident).
#define UPB_ROWS 8
#define UPB_MAXT 8
#define UPB_THREADS 320
#define UPB_NC 4   // columns (k, j) per thread: K * w <= UPB_NC * UPB_THREADS

struct UpBwdArgs {
    const float *gp, *p;
    float *T, *out;
    int B, K, d, h, w, D, H, W;
    float sd, sh, sw;
    int softmax, nsplit, dbg;
    unsigned ldlr;     // channel stride of the low-resolution gradient (padding zeroed by upsoftmax_bwd_d_kernel)
    // LG form: the gradient of the probabilities is not read but evaluated from the loss coefficients (loss_bwd_kernel's expression)
    const uint8_t *lab;
    const float *coef, *gscale;
};

// EXACT: K == KMAX and softmax at compile time (no per-class branches; the 4-class head of every BraTS configuration)
template <int KMAX, int NC, int NI, bool LG = false, bool EXACT = false>   // NI: (row, quad) items per thread, UPB_ROWS * W / 4 <= NI * UPB_THREADS
__global__ __launch_bounds__(UPB_THREADS) void upsoftmax_bwd_plane_kernel(UpBwdArgs a_) {
    extern __shared__ float stage[];   // [2][UPB_ROWS][K][W]
    UpBwdArgs a = a_;
    if constexpr (EXACT) { a.K = KMAX; a.softmax = 1; a.dbg = 0; }
    const int tid = threadIdx.x;
    if ((a.dbg & 0xe0) == 0x20) return;
    const int K = a.K, W = a.W, H = a.H, w = a.w, h = a.h;
    const int split = blockIdx.x % a.nsplit, bz = blockIdx.x / a.nsplit;
    const int b = bz / a.D, z = bz - b * a.D;
    const size_t V = (size_t)a.D * H * W;
    // band of low-res rows [ra, rb) of this workgroup and the high-res rows that touch it
    const int ra = (int)((long long)h * split / a.nsplit), rb = (int)((long long)h * (split + 1) / a.nsplit);
    // superset of the contributing rows without search loops: rows outside add weight 0 to the band
    int ha = (int)floorf(((float)(ra - 1) + 0.5f) / a.sh - 0.5f) - 1;
    int hb = (int)ceilf(((float)rb + 0.5f) / a.sh - 0.5f) + 1;
    ha = ha < 0 ? 0 : ha;
    hb = hb > H - 1 ? H - 1 : hb;
    // columns (k, j) owned by this thread and their W-axis taps: UPB_MAXT consecutive high-res columns
    // starting one below the first possible contributor (the host checks 2 W / w + 3 <= UPB_MAXT)
    int tap0[NC];
    float tapw[NC][UPB_MAXT];
    float a0[NC], a1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = tid + c * UPB_THREADS;
        a0[c] = a1[c] = 0.f;
        const int j = col % w;
        int x0 = (int)floorf(((float)(j - 1) + 0.5f) / a.sw - 0.5f) - 1;
        x0 = x0 < 0 ? 0 : (x0 > W - UPB_MAXT ? (W - UPB_MAXT > 0 ? W - UPB_MAXT : 0) : x0);
        x0 &= ~1;      // even first tap (W % 4 == 0, the window has one spare tap): the fold reads its taps as 8-byte pairs, conflict-free
        tap0[c] = x0;
#pragma unroll
        for (int t = 0; t < UPB_MAXT; ++t) tapw[c][t] = (x0 + t < W && col < K * w) ? lin_weight(lin_coord(x0 + t, a.sw, w), j) : 0.f;
    }
    if ((a.dbg & 0xe0) == 0x40) { if (tapw[0][0] == 123.f) a.T[0] = 1.f; return; }
    const int QW = W / 4;                          // float4 groups per row (W % 4 == 0)
    const int nitem = UPB_ROWS * QW;               // (row, quad) items per batch
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 gq[NI][KMAX], pq[NI][KMAX];
    unsigned labq[NI];
    float al[KMAX], be[KMAX], ga[KMAX];
    if constexpr (LG) {
        const float gs = a.gscale ? *a.gscale : 1.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            al[k] = be[k] = ga[k] = 0.f;
            if (k < K) {
                al[k] = gs * a.coef[((size_t)b * K + k) * 4 + 1];
                be[k] = gs * a.coef[((size_t)b * K + k) * 4 + 2];
                ga[k] = gs * a.coef[((size_t)b * K + k) * 4 + 3];
            }
        }
    }
    auto fetch = [&](int hbase) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = tid + it * UPB_THREADS;
            const int r = item / QW, q = item - r * QW;
            const int hh = hbase + r;
            if (item >= nitem || (a.dbg & 2)) continue;
            const size_t off = ((size_t)z * H + (hh <= hb ? hh : ha)) * W + 4 * q;
            if constexpr (LG) labq[it] = *reinterpret_cast<const unsigned *>(a.lab + (size_t)b * V + off);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    if constexpr (!LG) gq[it][k] = *reinterpret_cast<const f4 *>(a.gp + ((size_t)b * K + k) * V + off);
                    if (a.softmax || LG) pq[it][k] = *reinterpret_cast<const f4 *>(a.p + ((size_t)b * K + k) * V + off);
                }
            }
        }
    };
    int cur = ra;   // low-res row held in a0 (a1 holds cur + 1)
    float *Tb = a.T + (((size_t)b * K) * a.D + z) * (size_t)h * w;   // + k * D*h*w
    const size_t Tk = (size_t)a.D * h * w;
    fetch(ha);
    int buf = 0;
    for (int hbase = ha; hbase <= hb; hbase += UPB_ROWS, buf ^= 1) {
        float *st = stage + (size_t)buf * UPB_ROWS * K * W;
        // ---- phase 1: softmax gradient of the batch -> LDS
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = tid + it * UPB_THREADS;
            if (item < nitem) {
                const int r = item / QW, q = item - r * QW;
                if constexpr (LG) {     // d loss / d p = alpha t + beta p + gamma, as loss_bwd_kernel writes it
                    const int l0 = labq[it] & 255, l1 = (labq[it] >> 8) & 255, l2 = (labq[it] >> 16) & 255, l3 = labq[it] >> 24;
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
                        if (k < K) {
                            gq[it][k].x = (l0 == k ? al[k] : 0.f) + be[k] * pq[it][k].x + ga[k];
                            gq[it][k].y = (l1 == k ? al[k] : 0.f) + be[k] * pq[it][k].y + ga[k];
                            gq[it][k].z = (l2 == k ? al[k] : 0.f) + be[k] * pq[it][k].z + ga[k];
                            gq[it][k].w = (l3 == k ? al[k] : 0.f) + be[k] * pq[it][k].w + ga[k];
                        }
                }
                f4 dot = {0.f, 0.f, 0.f, 0.f};
                if (a.softmax) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
                        if (k < K) dot += pq[it][k] * gq[it][k];
                }
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K) {
                        const f4 v = a.softmax ? pq[it][k] * (gq[it][k] - dot) : gq[it][k];
                        *reinterpret_cast<f4 *>(st + ((size_t)r * K + k) * W + 4 * q) = v;
                    }
            }
        }
        if (hbase + UPB_ROWS <= hb) fetch(hbase + UPB_ROWS);
        __syncthreads();
        // ---- phase 2: W fold (taps) and H fold (sliding window), rows in order
        for (int r = 0; r < UPB_ROWS; ++r) {
            const int hh = hbase + r;
            if (hh > hb || (a.dbg & 1)) break;
            const Lin ly = lin_coord(hh, a.sh, h);
            // rows below i0 are complete: emit them and slide the window (uniform over the workgroup)
            while (cur < ly.i0 && cur < rb) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int col = tid + c * UPB_THREADS;
                    if (col < K * w) {
                        const int k = col / w, j = col - k * w;
                        Tb[k * Tk + (size_t)cur * w + j] = a0[c];
                        a0[c] = a1[c];
                        a1[c] = 0.f;
                    }
                }
                ++cur;
            }
            const float *sr = st + (size_t)r * K * W;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = tid + c * UPB_THREADS;
                if (col < K * w) {
                    const int k = col / w;
                    float v = 0.f;
                    // lanes own consecutive low-res columns, i.e. taps two floats apart: single-float reads were 2-way bank conflicts
                    // on every tap; tap0 is even and tap0 + UPB_MAXT <= W (clamped above, W >= UPB_MAXT), so pairs are aligned and in range
                    const float2 *sp = reinterpret_cast<const float2 *>(sr + k * W + tap0[c]);
#pragma unroll
                    for (int t = 0; t < UPB_MAXT; t += 2) {
                        const float2 pr = sp[t >> 1];
                        v += tapw[c][t] * pr.x;
                        v += tapw[c][t + 1] * pr.y;
                    }
                    const float c0 = (ly.i0 == cur ? ly.w0 : 0.f) + (ly.i1 == cur ? ly.w1 : 0.f);
                    const float c1 = (ly.i0 == cur + 1 ? ly.w0 : 0.f) + (ly.i1 == cur + 1 ? ly.w1 : 0.f);
                    a0[c] += c0 * v;
                    a1[c] += c1 * v;
                }
            }
        }
    }
    if ((a.dbg & 0xe0) == 0x60) { if (a0[0] == 123.f) a.T[0] = 1.f; return; }
    // flush the window
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = tid + c * UPB_THREADS;
        if (col < K * w) {
            const int k = col / w, j = col - k * w;
            if (cur < rb) Tb[k * Tk + (size_t)cur * w + j] = a0[c];
            if (cur + 1 < rb) Tb[k * Tk + (size_t)(cur + 1) * w + j] = a1[c];
            for (int i = cur + 2; i < rb; ++i) Tb[k * Tk + (size_t)i * w + j] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void upsoftmax_bwd_d_kernel(UpBwdArgs a) {
    const int hw = a.h * a.w;
    const int iz = blockIdx.y, bk = blockIdx.z;
    int z0, z1;
    contrib_range(iz, a.sd, a.d, a.D, z0, z1);
    float wz[UPB_MAXT];
#pragma unroll
    for (int t = 0; t < UPB_MAXT; ++t) wz[t] = z0 + t <= z1 ? lin_weight(lin_coord(z0 + t, a.sd, a.d), iz) : 0.f;
    const float *Tb = a.T + (size_t)bk * a.D * hw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < UPB_MAXT; ++t) {
            const int z = z0 + t <= z1 ? z0 + t : z1;   // weight 0 beyond the range
            acc += wz[t] * Tb[(size_t)z * hw + i];
        }
        a.out[(size_t)bk * a.ldlr + (size_t)iz * hw + i] = acc;
    }
    if (iz == 0 && blockIdx.x == 0 && a.ldlr > (unsigned)(a.d * hw)) {     // channel-padded gradient: the padding is zero (ops.chan_stride)
        const unsigned npad = a.ldlr - (unsigned)(a.d * hw);
        if (threadIdx.x < npad) a.out[(size_t)bk * a.ldlr + (size_t)a.d * hw + threadIdx.x] = 0.f;
    }
}

// ------------------------------------------------------------------------------ losses
// stats[b][k] = {sum p, sum p^2, sum p*t, sum t} in double (one pass, fp64 accumulation).
template <int KMAX>
__global__ __launch_bounds__(256) void loss_stats_kernel(const float *__restrict__ p, const uint8_t *__restrict__ lab,
                                                        double *stats, int K, long long V) {
    const int b = blockIdx.y;
    double sp[KMAX], sp2[KMAX], spt[KMAX], st[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) sp[k] = sp2[k] = spt[k] = st[k] = 0.0;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long long)gridDim.x * 256) {
        const int l = lab[(size_t)b * V + v];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const double pv = (double)p[((size_t)b * K + k) * V + v];
                sp[k] += pv;
                sp2[k] += pv * pv;
                if (l == k) {
                    spt[k] += pv;
                    st[k] += 1.0;
                }
            }
    }
    __shared__ double red[4][KMAX * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        double v4[4] = {sp[k], sp2[k], spt[k], st[k]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = v4[j];
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) red[wave][k * 4 + j] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < K * 4) {
        const double s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(&stats[(size_t)b * K * 4 + threadIdx.x], s);
    }
}

// Vectorised form for V % 4 == 0: 16-byte loads of the probabilities, 4 labels per 32-bit load, fp32 sums of each
// quad promoted to the fp64 accumulators, and FEW workgroups (128 per sample): the first version launched 2 048
// workgroups whose 32 k same-address double atomics dominated its 46 us.
// PART: every workgroup stores its sums to its own row of `stats` ([b][workgroup][K * 4]) instead of adding them to shared
// accumulators: no same-address double atomics (256 per address and launch serialise in the L2), no clear kernel, and a loss that
// is bit-reproducible; loss_finalize_kernel adds the rows in a fixed order.
template <int KMAX, bool PART = false>
__global__ __launch_bounds__(256) void loss_stats_vec_kernel(const float *__restrict__ p, const uint8_t *__restrict__ lab,
                                                            double *stats, int K, long long V) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y;
    double sp[KMAX], sp2[KMAX], spt[KMAX], st[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) sp[k] = sp2[k] = spt[k] = st[k] = 0.0;
    const long long nq = V / 4;
    const unsigned *lab4 = reinterpret_cast<const unsigned *>(lab + (size_t)b * V);
    for (long long qd = (long long)blockIdx.x * 256 + threadIdx.x; qd < nq; qd += (long long)gridDim.x * 256) {
        const unsigned l4 = lab4[qd];
        const int l0 = l4 & 255, l1 = (l4 >> 8) & 255, l2 = (l4 >> 16) & 255, l3 = l4 >> 24;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const f4 v = *reinterpret_cast<const f4 *>(p + ((size_t)b * K + k) * V + 4 * qd);
                const float t0 = l0 == k ? 1.f : 0.f, t1 = l1 == k ? 1.f : 0.f, t2 = l2 == k ? 1.f : 0.f, t3 = l3 == k ? 1.f : 0.f;
                sp[k] += (double)((v.x + v.y) + (v.z + v.w));
                sp2[k] += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
                spt[k] += (double)((t0 * v.x + t1 * v.y) + (t2 * v.z + t3 * v.w));
                st[k] += (double)((t0 + t1) + (t2 + t3));
            }
    }
    __shared__ double red[4][KMAX * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        double v4[4] = {sp[k], sp2[k], spt[k], st[k]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double sv = v4[j];
            for (int off = 32; off >= 1; off >>= 1) sv += __shfl_xor(sv, off);
            if (lane == 0) red[wave][k * 4 + j] = sv;
        }
    }
    __syncthreads();
    if (threadIdx.x < K * 4) {
        const double sv = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (PART) stats[((size_t)b * gridDim.x + blockIdx.x) * K * 4 + threadIdx.x] = sv;
        else atomicAdd(&stats[(size_t)b * K * 4 + threadIdx.x], sv);
    }
}

// coef[b][k] = {value, alpha, beta, gamma} with  dloss/dp[b,k,v] = alpha * t_v + beta * p_v + gamma
__global__ void loss_finalize_kernel(const double *stats, float *coef, float *loss, int B, int K, long long V, int kind,
                                     float param, const double *rows = nullptr, int nrows = 0, double *stats_out = nullptr) {
    __shared__ double part[1024];
    if (rows) {   // per-workgroup rows [b][nrows][K * 4] -> stats_out[b][K * 4], every sum in the same fixed order on every run:
                  // n = K * 4 consecutive threads read one row (coalesced), G = 256 / n rows per pass; the G partial sums of a
                  // statistic are then added in order
        const int n = K * 4, G = (int)blockDim.x / n;
        const int g = threadIdx.x / n, i = threadIdx.x - g * n;
        if (B > 1 && B <= G) {     // all samples in ONE pass: G / B groups of n threads per sample (round 4b: the loop over samples below
                                   // paid a global round trip and two barriers per sample)
            const int Gb = G / B, b = g / Gb, gg = g - b * Gb;
            double acc = 0.0;
            if (b < B) {
                const double *rb = rows + (size_t)b * nrows * n + i;
                int w = gg;
                for (; w + 3 * Gb < nrows; w += 4 * Gb) {
                    double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = rb[(size_t)(w + u * Gb) * n];
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc += v[u];
                }
                for (; w < nrows; w += Gb) acc += rb[(size_t)w * n];
            }
            part[threadIdx.x] = acc;
            __syncthreads();
            if (b < B && gg == 0) {
                double t = 0.0;
                for (int u = 0; u < Gb; ++u) t += part[(b * Gb + u) * n + i];
                stats_out[b * n + i] = t;
            }
            __syncthreads();
        } else
        for (int b = 0; b < B; ++b) {
            double acc = 0.0;
            if (g < G) {
                const double *rb = rows + (size_t)b * nrows * n + i;
                int w = g;
                for (; w + 7 * G < nrows; w += 8 * G) {     // eight independent loads in flight, added in row order
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = rb[(size_t)(w + u * G) * n];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += v[u];
                }
                for (; w < nrows; w += G) acc += rb[(size_t)w * n];
            }
            part[threadIdx.x] = acc;
            __syncthreads();
            if ((int)threadIdx.x < n) {
                double t = part[threadIdx.x];
                for (int u = 1; u < G; ++u) t += part[u * n + threadIdx.x];
                stats_out[b * n + threadIdx.x] = t;
            }
            __syncthreads();
        }
        stats = stats_out;
    }
    double local = 0.0;
    const double n = (double)V, inv_bk = 1.0 / ((double)B * K);
    for (int i = threadIdx.x; i < B * K; i += blockDim.x) {
        const double sp = stats[i * 4 + 0], sp2 = stats[i * 4 + 1], spt = stats[i * 4 + 2], st = stats[i * 4 + 3];
        double value, alpha, beta, gamma, term;
        if (kind == 0) {  // Pearson (custom_losses.py:32-39): centred sums, eps inside the sqrt
            const double pm = sp / n, tm = st / n;
            const double tp = spt - n * pm * tm, tt = st - n * tm * tm, pp = sp2 - n * pm * pm;
            const double den = sqrt(tt * pp + 1e-7);
            value = tp / den;
            // dr/dp_v = (t_v - tm)/den - tp*tt*(p_v - pm)/den^3 ; loss = mean(0.5 - 0.5 r)
            const double c1 = 1.0 / den, c2 = -tp * tt / (den * den * den);
            const double dl = -0.5 * inv_bk;
            alpha = dl * c1;
            beta = dl * c2;
            gamma = dl * (-c1 * tm - c2 * pm);
            term = 1.0 - (value + 1.0) * 0.5;
        } else {  // soft Dice (custom_losses.py:88-90)
            const double uni = st + sp + 1e-7;
            value = 2.0 * spt / uni;
            double dl;  // dloss/dvalue for this (b,k)
            if (kind == 1) {
                dl = -inv_bk;
                term = 1.0 - value;
            } else {  // ExpDice: mean((-ln clamp(d))^e)
                const double lo = 1e-7, hi = 1.0 - 1e-7;
                const double dc = value < lo ? lo : (value > hi ? hi : value);
                const double nl = -log(dc);
                term = pow(nl, (double)param);
                const bool inside = value >= lo && value <= hi;
                dl = inside ? inv_bk * (double)param * pow(nl, (double)param - 1.0) * (-1.0 / dc) : 0.0;
            }
            alpha = dl * 2.0 / uni;
            beta = 0.0;
            gamma = dl * (-2.0 * spt / (uni * uni));
        }
        coef[i * 4 + 0] = (float)value;
        coef[i * 4 + 1] = (float)alpha;
        coef[i * 4 + 2] = (float)beta;
        coef[i * 4 + 3] = (float)gamma;
        local += term;
    }
    part[threadIdx.x] = local;
    __syncthreads();
    if (threadIdx.x == 0) {       // at most min(B K, blockDim) threads hold a term: added in index order (a 10-level tree of barriers before)
        const int nt = B * K < (int)blockDim.x ? B * K : (int)blockDim.x;
        double t = 0.0;
        for (int j = 0; j < nt; ++j) t += part[j];
        *loss = (float)(t * inv_bk);
    }
}

template <int KMAX>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float *__restrict__ p, const uint8_t *__restrict__ lab,
                                                      const float *__restrict__ coef, const float *__restrict__ gscale,
                                                      float *__restrict__ g, int K, long long V) {
    const int b = blockIdx.y;
    const float gs = gscale ? *gscale : 1.f;
    float al[KMAX], be[KMAX], ga[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        al[k] = be[k] = ga[k] = 0.f;
        if (k < K) {
            al[k] = gs * coef[((size_t)b * K + k) * 4 + 1];
            be[k] = gs * coef[((size_t)b * K + k) * 4 + 2];
            ga[k] = gs * coef[((size_t)b * K + k) * 4 + 3];
        }
    }
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long long)gridDim.x * 256) {
        const int l = lab[(size_t)b * V + v];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const size_t idx = ((size_t)b * K + k) * V + v;
                g[idx] = (l == k ? al[k] : 0.f) + be[k] * p[idx] + ga[k];
            }
    }
}

__global__ __launch_bounds__(256) void labels_kernel(const float *__restrict__ lf, const int *rfrom, const int *rto, int nmap,
                                                    uint8_t *lu, float *onehot, int K, long long V, int B) {
    const size_t total = (size_t)V * B;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int orig = (int)lf[idx];  // y.to(dtype=int) truncates (experiments/utils.py:86)
        int l = orig;
        for (int j = 0; j < nmap; ++j)
            if (orig == rfrom[j]) l = rto[j];  // every key matched against the ORIGINAL label
        if (lu) lu[idx] = (uint8_t)l;
        if (onehot) {
            const size_t b = idx / V, v = idx % V;
            for (int k = 0; k < K; ++k) onehot[(b * K + k) * V + v] = (l == k) ? 1.f : 0.f;
        }
    }
}

// class map only, four voxels per thread (16-byte loads, one 32-bit store): the per-voxel form above moves 21 MB in 8.4 us
__global__ __launch_bounds__(256) void labels4_kernel(const float *__restrict__ lf, const int *rfrom, const int *rto, int nmap,
                                                     unsigned *__restrict__ lu4, size_t nquad) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nquad; q += (size_t)gridDim.x * 256) {
        const float4 f = reinterpret_cast<const float4 *>(lf)[q];
        const int orig[4] = {(int)f.x, (int)f.y, (int)f.z, (int)f.w};     // y.to(dtype=int) truncates (experiments/utils.py:86)
        unsigned out = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int l = orig[e];
            for (int j = 0; j < nmap; ++j)
                if (orig[e] == rfrom[j]) l = rto[j];  // every key matched against the ORIGINAL label
            out |= (unsigned)(l & 255) << (8 * e);
        }
        lu4[q] = out;
    }
}

__global__ __launch_bounds__(256) void eltwise_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                     float *__restrict__ out, size_t n, int op, int act,
                                                     float alpha = 1.f, float beta = 1.f) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v;
        if (op == 0) v = act_apply(a[i], act);                       // y = act(x)
        else if (op == 1) v = a[i] * act_grad_from_out(b[i], act);   // gx = g * act'(y)
        else if (op == 2) v = a[i] + b[i];
        else v = b ? fmaf(alpha, a[i], beta * b[i]) : alpha * a[i];  // out = alpha a + beta b
        out[i] = v;
    }
}

// y[b][c][v] = act(y[b][c][v] + bias[c]) in place (epilogue of the GEMM-routed 1x1x1 convolutions)
__global__ __launch_bounds__(256) void bias_act_kernel(float *__restrict__ y, const float *__restrict__ bias, int C, long long V, int act) {
    const int bc = blockIdx.y;
    const float bv = bias ? bias[bc % C] : 0.f;
    float *p = y + (size_t)bc * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) p[i] = act_apply(p[i] + bv, act);
}

static int grid1d(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace hno

using namespace hno;

// kernels templated on the class-count bound: 4, 8, 16 or 32 registers per voxel
#define HNO_KSEL(K, X) do { if ((K) <= 4) { X(4); } else if ((K) <= 8) { X(8); } else if ((K) <= 16) { X(16); } else { X(32); } } while (0)

static int up_fill(UpArgs &a, int B, int K, int d, int h, int w, int D, int H, int W, int softmax) {
    HNO_REQUIRE(B > 0 && K > 0 && d > 0 && h > 0 && w > 0 && D > 0 && H > 0 && W > 0, "hno_upsoftmax: bad size");
    // (round 6: up to 32 classes -- the reference takes any out_channels; more than 8 run the plain voxel-form kernels, the row / separable
    // forms of the 2 ... 8-class heads stay as they are)
    if (K > 32) return fail(HNO_ELIMIT, "hno_upsoftmax: K=%d output channels (max 32)", K);
    a.B = B; a.K = K; a.d = d; a.h = h; a.w = w; a.D = D; a.H = H; a.W = W;
    a.sd = (float)d / (float)D; a.sh = (float)h / (float)H; a.sw = (float)w / (float)W;
    a.softmax = softmax;
    a.ldlr = (unsigned)((long long)d * h * w);
    return HNO_OK;
}

static int up_set_ld(UpArgs &a, long long ldlr) {
    const long long v = (long long)a.d * a.h * a.w;
    if (ldlr == 0) ldlr = v;
    HNO_REQUIRE(ldlr >= v && ldlr < v + 64 && ldlr < (1ll << 31), "hno_upsoftmax: channel stride %lld for %lld low-resolution voxels", ldlr, v);
    a.ldlr = (unsigned)ldlr;
    return HNO_OK;
}

static bool uprows_ok(int K, int d, int h, int w, int D, int H, int W) {
    (void)d; (void)h; (void)D; (void)H;
    return K <= 8 && W % 2 == 0 && W <= 128 && w <= 127 && w >= 2;   // (w = 128 would need the duplicated column at index 128)
}

template <bool STATS>
static void uprows_launch(const UpArgs &a, int gx, int B, hipStream_t s, const uint8_t *labels, double *rows) {
    const dim3 g(gx, B), blk(64 * UPR_WAVES);
    switch (a.K) {
        case 2: hipLaunchKernelGGL((uphead_rows_kernel<2, STATS, true>), g, blk, 0, s, a, labels, rows); break;
        case 3: hipLaunchKernelGGL((uphead_rows_kernel<3, STATS, true>), g, blk, 0, s, a, labels, rows); break;
        case 4: hipLaunchKernelGGL((uphead_rows_kernel<4, STATS, true>), g, blk, 0, s, a, labels, rows); break;
        case 1: hipLaunchKernelGGL((uphead_rows_kernel<4, STATS, false>), g, blk, 0, s, a, labels, rows); break;
        default: hipLaunchKernelGGL((uphead_rows_kernel<8, STATS, false>), g, blk, 0, s, a, labels, rows); break;
    }
}

// workgroups per sample of the row kernel (= rows of the statistics workspace per sample, <= 1 024): two 8-wave workgroups per CU
// are resident (registers), so the grid is 512 workgroups when there are rows enough (HNO_UPR_WGS overrides: A/B); rows_per_wave -> the longest fp32 run of a lane
static int uprows_plan(int B, int D, int H, bool stats, int &rows_per_wave) {
    const long long nrow = (long long)D * H;
    static const int env = getenv("HNO_UPR_WGS") ? atoi(getenv("HNO_UPR_WGS")) : 0;
    const int chip = env >= 64 && env <= 8192 ? env : 512;
    long long gx = (chip + B - 1) / B;
    const long long most = (nrow + UPR_WAVES - 1) / UPR_WAVES;       // at least one row per wave
    if (gx > most) gx = most;
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    (void)stats;
    rows_per_wave = (int)((nrow + gx * UPR_WAVES - 1) / (gx * UPR_WAVES)) + 1;
    return (int)gx;
}

// ldlr: floats between consecutive (b, k) volumes of logits_lr (0 = d h w; a channel-padded stride: ops.chan_stride)
extern "C" int hno_upsoftmax_fwd_ld(const float *logits_lr, float *probs, int B, int K, int d, int h, int w,
                                    int D, int H, int W, int softmax, long long ldlr, void *stream) {
    HNO_REQUIRE(logits_lr && probs, "hno_upsoftmax_fwd: null pointer");
    UpArgs a = {};
    int rc = up_fill(a, B, K, d, h, w, D, H, W, softmax);
    if (rc) return rc;
    rc = up_set_ld(a, ldlr);
    if (rc) return rc;
    a.lr = logits_lr; a.out = probs;
    ProfScope _ps(KID_UPSOFTMAX_FWD, (hipStream_t)stream, 4.0 * B * K * ((double)d * h * w + (double)D * H * W));
    const long long items = (long long)B * D * H * ((W + 63) / 64);
    static const bool rows_off = getenv("HNO_HEAD_ROWS") && atoi(getenv("HNO_HEAD_ROWS")) == 0;      // A/B: the voxel-form kernel
    if (uprows_ok(K, d, h, w, D, H, W) && (long long)D * H * W < (1ll << 31) && !(debug_flags() & 16) && !rows_off) {
        int rpw;
        const int gx = uprows_plan(B, D, H, false, rpw);
        uprows_launch<false>(a, gx, B, (hipStream_t)stream, nullptr, nullptr);
    } else if (K <= 8 && items < (1ll << 31) && (long long)d * h * w < (1ll << 31) && !(debug_flags() & 16)) {
        long long grid = (items + 3) / 4;
        if (grid > 8192) grid = 8192;
        if (debug_grid()) grid = debug_grid();
        if (K <= 4) hipLaunchKernelGGL((uphead_seg_kernel<4, false>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, nullptr);
        else hipLaunchKernelGGL((uphead_seg_kernel<8, false>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, nullptr);
    } else {
        const int grid = grid1d((size_t)B * D * H * W);
#define X(KM) hipLaunchKernelGGL(upsoftmax_fwd_kernel<KM>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a)
        HNO_KSEL(K, X);
#undef X
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_uphead_loss_supported(int B, int K, int d, int h, int w, int D, int H, int W) {
    if (B <= 0 || !uprows_ok(K, d, h, w, D, H, W) || (long long)D * H * W >= (1ll << 31)) return 0;
    int rpw;
    uprows_plan(B, D, H, true, rpw);
    return rpw <= 64 && ((long long)D * H * W) % 4 == 0;       // fp32 sums of <= 128 probabilities per lane before the fp64 stages
}

extern "C" size_t hno_uphead_loss_workspace_doubles(int B, int K) {
    if (B <= 0 || K <= 0) return 0;
    return (size_t)B * K * 4 * (1 + 1024);
}

// Softmax head and loss statistics in one pass (round 4): trilinear upsampling of the low-resolution logits + softmax
// (nets/hnosegxs.py:174-180) -> probs, and from the same registers the sums PCCLoss / DiceLoss / ExpDiceLoss need
// (nets/custom_losses.py:17-133): coef[b][k] = {value, alpha, beta, gamma}, *loss as hno_loss_fwd_ws leaves them.
extern "C" int hno_uphead_loss_fwd(const float *logits_lr, const unsigned char *labels, float *probs, double *workspace,
                                   size_t workspace_doubles, float *coef, float *loss, int B, int K, int d, int h, int w, int D, int H, int W,
                                   long long ldlr, int kind, float param, void *stream) {
    HNO_REQUIRE(logits_lr && labels && probs && workspace && coef && loss, "hno_uphead_loss_fwd: null pointer");
    HNO_REQUIRE(kind >= 0 && kind <= 2, "hno_uphead_loss_fwd: kind must be 0 (PCC), 1 (Dice) or 2 (ExpDice)");
    HNO_REQUIRE(((size_t)labels & 1) == 0, "hno_uphead_loss_fwd: labels must be 2-byte aligned");
    if (!hno_uphead_loss_supported(B, K, d, h, w, D, H, W)) return fail(HNO_ELIMIT, "hno_uphead_loss_fwd: shape not covered");
    HNO_REQUIRE(workspace_doubles >= hno_uphead_loss_workspace_doubles(B, K), "hno_uphead_loss_fwd: workspace too small");
    UpArgs a = {};
    int rc = up_fill(a, B, K, d, h, w, D, H, W, 1);
    if (rc) return rc;
    rc = up_set_ld(a, ldlr);
    if (rc) return rc;
    a.lr = logits_lr; a.out = probs;
    hipStream_t s = (hipStream_t)stream;
    int rpw;
    const int gx = uprows_plan(B, D, H, true, rpw);
    double *rows = workspace + (size_t)B * K * 4;
    const long long V = (long long)D * H * W;
    {
        ProfScope _ps(KID_UPSOFTMAX_FWD, s, 4.0 * B * K * ((double)d * h * w + (double)V) + (double)B * V);
        uprows_launch<true>(a, gx, B, s, labels, rows);
    }
    HNO_CHECK_LAUNCH();
    {
        ProfScope _ps2(KID_LOSS_FINALIZE, s);
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, s, (const double *)workspace, coef, loss, B, K, V, kind, param,
                           (const double *)rows, gx, workspace);
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_upsoftmax_fwd(const float *logits_lr, float *probs, int B, int K, int d, int h, int w,
                                 int D, int H, int W, int softmax, void *stream) {
    return hno_upsoftmax_fwd_ld(logits_lr, probs, B, K, d, h, w, D, H, W, softmax, 0, stream);
}

extern "C" int hno_up_argmax_ld(const float *logits_lr, unsigned char *labels, int B, int K, int d, int h, int w, int D, int H,
                                int W, long long ldlr, void *stream) {
    HNO_REQUIRE(logits_lr && labels, "hno_up_argmax: null pointer");
    UpArgs a = {};
    int rc = up_fill(a, B, K, d, h, w, D, H, W, 0);
    if (rc) return rc;
    rc = up_set_ld(a, ldlr);
    if (rc) return rc;
    a.lr = logits_lr;
    ProfScope _ps(KID_UPSOFTMAX_FWD, (hipStream_t)stream, 4.0 * B * K * (double)d * h * w + (double)B * D * H * W);
    const long long items = (long long)B * D * H * ((W + 63) / 64);
    if (K <= 8 && items < (1ll << 31) && (long long)d * h * w < (1ll << 31) && !(debug_flags() & 16)) {
        long long grid = (items + 3) / 4;
        if (grid > 8192) grid = 8192;
        if (debug_grid()) grid = debug_grid();
        if (K <= 4) hipLaunchKernelGGL((uphead_seg_kernel<4, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, labels);
        else hipLaunchKernelGGL((uphead_seg_kernel<8, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, labels);
    } else {
        const int grid = grid1d((size_t)B * D * H * W);
#define X(KM) hipLaunchKernelGGL(upargmax_kernel<KM>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, labels)
        HNO_KSEL(K, X);
#undef X
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_up_argmax(const float *logits_lr, unsigned char *labels, int B, int K, int d, int h, int w, int D, int H,
                             int W, void *stream) {
    return hno_up_argmax_ld(logits_lr, labels, B, K, d, h, w, D, H, W, 0, stream);
}

static bool upb_separable_ok(int K, int d, int h, int w, int D, int H, int W) {
    if (K > 8) return false;
    if (W % 4 != 0 || W > 4 * UPB_THREADS / UPB_ROWS * 2 || K * w > UPB_NC * UPB_THREADS) return false;
    if (w > W || h > H || d > D) return false;                       // taps bound assumes upsampling
    return (int)ceil(2.0 * W / w) + 3 <= UPB_MAXT && W >= UPB_MAXT;   // taps per low-res column (+ search margin)
}

extern "C" size_t hno_upsoftmax_bwd_workspace_bytes(int B, int K, int d, int h, int w, int D, int H, int W) {
    (void)d;
    if (B <= 0 || K <= 0 || !upb_separable_ok(K, d, h, w, D, H, W)) return 0;
    return sizeof(float) * (size_t)B * K * D * h * w;
}

// ldlr: floats between consecutive (b, k) volumes of g_lr (0 = d h w; channel-padded: the padding is zeroed)
static int upsoftmax_bwd_impl(const float *g_probs, const float *probs, float *g_lr, void *workspace, int B, int K, int d,
                              int h, int w, int D, int H, int W, int softmax, long long ldlr, void *stream, const uint8_t *lab,
                              const float *coef, const float *gscale) {
    const bool lg = lab != nullptr;
    HNO_REQUIRE((g_probs || lg) && g_lr && (probs || !(softmax || lg)), "hno_upsoftmax_bwd: null pointer");
    UpArgs a = {};
    int rc = up_fill(a, B, K, d, h, w, D, H, W, softmax);
    if (rc) return rc;
    rc = up_set_ld(a, ldlr);
    if (rc) return rc;
    if (a.ldlr != (unsigned)((long long)d * h * w) && !(workspace && upb_separable_ok(K, d, h, w, D, H, W)))
        return fail(HNO_ELIMIT, "hno_upsoftmax_bwd: a channel-padded gradient needs the separable form");
    a.gp = g_probs; a.p = probs; a.out = g_lr;
    hipStream_t s = (hipStream_t)stream;
    const double abytes = 4.0 * B * K * ((double)d * h * w + (softmax ? 2.0 : 1.0) * D * H * W);
    if (lg && !(workspace && upb_separable_ok(K, d, h, w, D, H, W)))
        return fail(HNO_ELIMIT, "hno_upsoftmax_loss_bwd: shape not covered by the separable form");
    if (workspace && upb_separable_ok(K, d, h, w, D, H, W) && (lg || !(debug_flags() & 16) || a.ldlr != (unsigned)((long long)d * h * w))) {
        UpBwdArgs u = {};
        u.lab = lab; u.coef = coef; u.gscale = gscale;
        u.gp = g_probs; u.p = probs; u.T = (float *)workspace; u.out = g_lr;
        u.B = B; u.K = K; u.d = d; u.h = h; u.w = w; u.D = D; u.H = H; u.W = W;
        u.sd = a.sd; u.sh = a.sh; u.sw = a.sw; u.softmax = softmax;
        u.ldlr = a.ldlr;
        const int planes = B * D;
        int nsplit = (1024 + planes - 1) / planes;
        if (nsplit > h / 8) nsplit = h / 8;
        if (debug_grid()) nsplit = debug_grid();
        if (nsplit < 1) nsplit = 1;
        u.nsplit = nsplit;
        const size_t lds = sizeof(float) * 2 * UPB_ROWS * (size_t)K * W;
        const int nc = (K * w + UPB_THREADS - 1) / UPB_THREADS;
        {
            ProfScope _ps(KID_UPSOFTMAX_BWD, s, abytes);
            const dim3 g(planes * nsplit), blk(UPB_THREADS);
            const bool ni1 = UPB_ROWS * (W / 4) <= UPB_THREADS;
            u.dbg = debug_flags();
            const bool exact = K == 4 && softmax && !u.dbg && nc <= 1 && ni1;
            if (exact && lg) hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<4, 1, 1, true, true>), g, blk, lds, s, u);
            else if (exact) hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<4, 1, 1, false, true>), g, blk, lds, s, u);
            else if (lg) {
                if (K <= 4 && nc <= 1 && ni1) hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<4, 1, 1, true>), g, blk, lds, s, u);
                else if (K <= 4) hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<4, UPB_NC, 2, true>), g, blk, lds, s, u);
                else if (nc <= 1 && ni1) hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<8, 1, 1, true>), g, blk, lds, s, u);
                else hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<8, UPB_NC, 2, true>), g, blk, lds, s, u);
            } else if (K <= 4 && nc <= 1 && ni1) hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<4, 1, 1>), g, blk, lds, s, u);
            else if (K <= 4) hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<4, UPB_NC, 2>), g, blk, lds, s, u);
            else if (nc <= 1 && ni1) hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<8, 1, 1>), g, blk, lds, s, u);
            else hipLaunchKernelGGL((upsoftmax_bwd_plane_kernel<8, UPB_NC, 2>), g, blk, lds, s, u);
        }
        HNO_CHECK_LAUNCH();
        {
            ProfScope _ps(KID_UPSOFTMAX_BWD_D, s, 4.0 * B * K * ((double)D + d) * h * w);
            int gx = (h * w + 2047) / 2048;   // ~8 elements per thread amortise the per-thread tap arithmetic
            if (gx > 64) gx = 64;
            hipLaunchKernelGGL(upsoftmax_bwd_d_kernel, dim3(gx, d, B * K), dim3(256), 0, s, u);
        }
        HNO_CHECK_LAUNCH();
        return HNO_OK;
    }
    const int grid = grid1d((size_t)B * d * h * w);
    {
        ProfScope _ps(KID_UPSOFTMAX_BWD, s, abytes);
#define X(KM) hipLaunchKernelGGL(upsoftmax_bwd_kernel<KM>, dim3(grid), dim3(256), 0, s, a)
        HNO_KSEL(K, X);
#undef X
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_upsoftmax_bwd_ld(const float *g_probs, const float *probs, float *g_lr, void *workspace, int B, int K, int d,
                                    int h, int w, int D, int H, int W, int softmax, long long ldlr, void *stream) {
    return upsoftmax_bwd_impl(g_probs, probs, g_lr, workspace, B, K, d, h, w, D, H, W, softmax, ldlr, stream, nullptr, nullptr, nullptr);
}

// Backward of (softmax head -> PCC / Dice / ExpDice loss) in one pass (round 4): d loss / d probs = gscale (alpha t + beta p + gamma) is
// evaluated from `coef` (what hno_loss_fwd / hno_uphead_loss_fwd left there) and the labels inside the head's backward instead of being
// written by hno_loss_bwd (67 MB per 2 x 4 x 128^3 batch) and read back.  Separable form only (hno_upsoftmax_bwd_workspace_bytes > 0),
// else HNO_ELIMIT: the caller then runs hno_loss_bwd + hno_upsoftmax_bwd_ld.
extern "C" int hno_upsoftmax_loss_bwd(const float *probs, const unsigned char *labels, const float *coef, const float *gscale, float *g_lr,
                                      void *workspace, int B, int K, int d, int h, int w, int D, int H, int W, long long ldlr, void *stream) {
    HNO_REQUIRE(probs && labels && coef && g_lr, "hno_upsoftmax_loss_bwd: null pointer");
    HNO_REQUIRE(((size_t)labels & 3) == 0, "hno_upsoftmax_loss_bwd: labels must be 4-byte aligned");
    return upsoftmax_bwd_impl(nullptr, probs, g_lr, workspace, B, K, d, h, w, D, H, W, 1, ldlr, stream, labels, coef, gscale);
}

extern "C" int hno_upsoftmax_bwd(const float *g_probs, const float *probs, float *g_lr, void *workspace, int B, int K, int d,
                                 int h, int w, int D, int H, int W, int softmax, void *stream) {
    return hno_upsoftmax_bwd_ld(g_probs, probs, g_lr, workspace, B, K, d, h, w, D, H, W, softmax, 0, stream);
}

extern "C" int hno_loss_fwd(const float *probs, const uint8_t *labels, double *stats, float *coef, float *loss,
                            int B, int K, long long V, int kind, float param, void *stream) {
    HNO_REQUIRE(probs && labels && stats && coef && loss && B > 0 && K > 0 && V > 0, "hno_loss_fwd: bad argument");
    HNO_REQUIRE(kind >= 0 && kind <= 2, "hno_loss_fwd: kind must be 0 (PCC), 1 (Dice) or 2 (ExpDice)");
    if (K > 32) return fail(HNO_ELIMIT, "hno_loss_fwd: K=%d classes (max 32)", K);
    hipStream_t s = (hipStream_t)stream;
    {   // accumulators of the statistics kernels (see clear_doubles: not a memset node)
        const int rc = clear_doubles(stats, B * K * 4, s);
        if (rc) return rc;
    }
    if (V % 4 == 0 && ((size_t)labels & 3) == 0 && ((size_t)probs & 15) == 0 && !(debug_flags() & 16)) {
        long long gq = (V / 4 + 255) / 256;
        if (gq > 256) gq = 256;   // measured (stats + finalize): 128 -> 37 us, 256 -> 29, 512 -> 32, 2048 -> 67 (atomic contention)
        if (debug_grid()) gq = debug_grid();
        ProfScope _ps(KID_LOSS_STATS, s, (double)B * V * (4.0 * K + 1));
#define X(KM) hipLaunchKernelGGL(loss_stats_vec_kernel<KM>, dim3((int)gq, B), dim3(256), 0, s, probs, labels, stats, K, V)
        HNO_KSEL(K, X);
#undef X
        HNO_CHECK_LAUNCH();
        { ProfScope _ps2(KID_LOSS_FINALIZE, s); hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, (const double *)stats, coef, loss, B, K, V, kind, param); }
        HNO_CHECK_LAUNCH();
        return HNO_OK;
    }
    long long gx = (V + 256 * 8 - 1) / (256 * 8);
    if (gx > 1024) gx = 1024;
    {
        ProfScope _ps(KID_LOSS_STATS, s, (double)B * V * (4.0 * K + 1));
#define X(KM) hipLaunchKernelGGL(loss_stats_kernel<KM>, dim3((int)gx, B), dim3(256), 0, s, probs, labels, stats, K, V)
        HNO_KSEL(K, X);
#undef X
    }
    HNO_CHECK_LAUNCH();
    { ProfScope _ps(KID_LOSS_FINALIZE, s); hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, (const double *)stats, coef, loss, B, K, V, kind, param); }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// the statistics through per-workgroup rows (no atomics, no clear kernel, bit-reproducible): workspace of hno_loss_workspace_doubles
// doubles; its first B K 4 doubles hold the reduced statistics afterwards (what hno_loss_fwd leaves in `stats`)
static int loss_rows(int B, long long V) {
    long long gq = (V / 4 + 255) / 256;
    static const int env = getenv("HNO_LOSS_ROWS") ? atoi(getenv("HNO_LOSS_ROWS")) : 0;
    const long long cap = (env > 0 && env <= 1024) ? env : 512;    // the workspace holds at most 1 024 rows per sample
    if (gq > cap) gq = cap;
    (void)B;
    return (int)gq;
}

extern "C" size_t hno_loss_workspace_doubles(int B, int K, long long V) {
    if (B <= 0 || K <= 0 || V <= 0) return 0;
    return (size_t)B * K * 4 * (1 + (size_t)(V % 4 == 0 ? 1024 : 0));      // room for up to 1 024 rows per sample
}

extern "C" int hno_loss_fwd_ws(const float *probs, const uint8_t *labels, double *workspace, size_t workspace_doubles, float *coef,
                               float *loss, int B, int K, long long V, int kind, float param, void *stream) {
    HNO_REQUIRE(probs && labels && workspace && coef && loss && B > 0 && K > 0 && V > 0, "hno_loss_fwd_ws: bad argument");
    HNO_REQUIRE(kind >= 0 && kind <= 2, "hno_loss_fwd_ws: kind must be 0 (PCC), 1 (Dice) or 2 (ExpDice)");
    if (K > 32) return fail(HNO_ELIMIT, "hno_loss_fwd_ws: K=%d classes (max 32)", K);
    const bool vec = V % 4 == 0 && ((size_t)labels & 3) == 0 && ((size_t)probs & 15) == 0 && !(debug_flags() & 16);
    int gq = loss_rows(B, V);
    if (gq > 1024) gq = 1024;
    if (!vec || workspace_doubles < (size_t)B * K * 4 * (1 + (size_t)gq))
        return hno_loss_fwd(probs, labels, workspace, coef, loss, B, K, V, kind, param, stream);      // the accumulating form
    hipStream_t s = (hipStream_t)stream;
    double *rows = workspace + (size_t)B * K * 4;
    {
        ProfScope _ps(KID_LOSS_STATS, s, (double)B * V * (4.0 * K + 1));
#define X(KM) hipLaunchKernelGGL((loss_stats_vec_kernel<KM, true>), dim3(gq, B), dim3(256), 0, s, probs, labels, rows, K, V)
        HNO_KSEL(K, X);
#undef X
    }
    HNO_CHECK_LAUNCH();
    {
        ProfScope _ps2(KID_LOSS_FINALIZE, s);
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, (const double *)workspace, coef, loss, B, K, V, kind, param,
                           (const double *)rows, gq, workspace);
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_loss_bwd(const float *probs, const uint8_t *labels, const float *coef, const float *gscale,
                            float *g_probs, int B, int K, long long V, void *stream) {
    HNO_REQUIRE(probs && labels && coef && g_probs && B > 0 && K > 0 && V > 0, "hno_loss_bwd: bad argument");
    if (K > 32) return fail(HNO_ELIMIT, "hno_loss_bwd: K=%d classes (max 32)", K);
    long long gx = (V + 256 * 4 - 1) / (256 * 4);
    if (gx > 2048) gx = 2048;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope _ps(KID_LOSS_BWD, s, (double)B * V * (8.0 * K + 1));
#define X(KM) hipLaunchKernelGGL(loss_bwd_kernel<KM>, dim3((int)gx, B), dim3(256), 0, s, probs, labels, coef, gscale, g_probs, K, V)
        HNO_KSEL(K, X);
#undef X
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_labels_prepare(const float *labels_f32, const int *remap_from, const int *remap_to, int n_remap,
                                  uint8_t *labels_u8, float *onehot, int B, int K, long long V, void *stream) {
    HNO_REQUIRE(labels_f32 && (labels_u8 || onehot) && B > 0 && V > 0, "hno_labels_prepare: bad argument");
    HNO_REQUIRE(n_remap == 0 || (remap_from && remap_to), "hno_labels_prepare: remap tables missing");
    const size_t total = (size_t)B * V;
    if (!onehot && total % 4 == 0 && ((size_t)labels_f32 & 15) == 0 && ((size_t)labels_u8 & 3) == 0) {
        ProfScope _ps(KID_LABELS, (hipStream_t)stream);
        hipLaunchKernelGGL(labels4_kernel, dim3(grid1d(total / 4)), dim3(256), 0, (hipStream_t)stream, labels_f32, remap_from, remap_to, n_remap,
                           (unsigned *)labels_u8, total / 4);
    } else {
        ProfScope _ps(KID_LABELS, (hipStream_t)stream);
        hipLaunchKernelGGL(labels_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, labels_f32, remap_from, remap_to, n_remap, labels_u8,
                           onehot, K, V, B);
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_act_fwd(const float *x, float *y, long long n, int act, void *stream) {
    HNO_REQUIRE(x && y && n > 0, "hno_act_fwd: bad argument");
    hipLaunchKernelGGL(eltwise_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream, x, (const float *)nullptr, y, (size_t)n, 0, act);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_act_bwd(const float *g, const float *y, float *gx, long long n, int act, void *stream) {
    HNO_REQUIRE(g && y && gx && n > 0, "hno_act_bwd: bad argument");
    hipLaunchKernelGGL(eltwise_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream, g, y, gx, (size_t)n, 1, act);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_bias_act(float *y, const float *bias, int B, int C, long long V, int act, void *stream) {
    HNO_REQUIRE(y && B > 0 && C > 0 && V > 0, "hno_bias_act: bad argument");
    if ((long long)B * C > 65535) return fail(HNO_ELIMIT, "hno_bias_act: B*C = %lld exceeds 65535", (long long)B * C);
    int gx = (int)((V + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(bias_act_kernel, dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, y, bias, C, V, act);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_add(const float *a, const float *b, float *out, long long n, void *stream) {
    HNO_REQUIRE(a && b && out && n > 0, "hno_add: bad argument");
    hipLaunchKernelGGL(eltwise_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)n, 2, 0);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// dst[r][v] = v < V ? src[r][v] : 0 for v in [0, ld_dst): moves (b, c) volumes between the contiguous layout (ld = V) and the
// channel-padded one (ld = V rounded up to a multiple of 32 floats, padding zeroed)
__global__ __launch_bounds__(256) void chan_restride_kernel(const float *__restrict__ src, float *__restrict__ dst, unsigned V,
                                                            unsigned ld_src, unsigned ld_dst) {
    const size_t r = blockIdx.y;
    const float *s = src + r * ld_src;
    float *d = dst + r * ld_dst;
    for (unsigned v = blockIdx.x * 256u + threadIdx.x; v < ld_dst; v += gridDim.x * 256u) d[v] = v < V ? s[v] : 0.f;
}

extern "C" int hno_chan_restride(const float *src, float *dst, long long rows, long long V, long long ld_src, long long ld_dst,
                                 void *stream) {
    HNO_REQUIRE(src && dst && rows > 0 && V > 0 && ld_src >= V && ld_dst >= V, "hno_chan_restride: bad argument");
    if (rows > 65535 || ld_dst >= (1ll << 32) || ld_src >= (1ll << 32))
        return fail(HNO_ELIMIT, "hno_chan_restride: %lld rows of %lld floats", rows, ld_dst);
    int gx = (int)((ld_dst + 1023) / 1024);
    if (gx > 128) gx = 128;
    hipLaunchKernelGGL(chan_restride_kernel, dim3(gx, (int)rows), dim3(256), 0, (hipStream_t)stream, src, dst, (unsigned)V,
                       (unsigned)ld_src, (unsigned)ld_dst);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// fp32 <-> bf16 over a flat extent (channel padding included): where an fp32 producer (the stem) meets the bf16 block chain of an
// autocast run and where the chain's bf16 output meets the fp32 head (round 6: bf16 activations in memory).  n % 4 == 0 is not needed.
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u)
        dst[i] = __builtin_bit_cast(unsigned short, (__bf16)src[i]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const unsigned short *__restrict__ src, float *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u)
        dst[i] = __builtin_bit_cast(float, (unsigned)src[i] << 16);
}
extern "C" int hno_cast_f32_bf16(const float *src, void *dst_bf16, long long n, void *stream) {
    HNO_REQUIRE(src && dst_bf16 && n > 0, "hno_cast_f32_bf16: bad argument");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream, src, (unsigned short *)dst_bf16, (size_t)n);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
extern "C" int hno_cast_bf16_f32(const void *src_bf16, float *dst, long long n, void *stream) {
    HNO_REQUIRE(src_bf16 && dst && n > 0, "hno_cast_bf16_f32: bad argument");
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)src_bf16, dst, (size_t)n);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_axpby(float alpha, const float *a, float beta, const float *b, float *out, long long n, void *stream) {
    HNO_REQUIRE(a && out && n > 0, "hno_axpby: bad argument");
    hipLaunchKernelGGL(eltwise_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)n, 3, 0, alpha, beta);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
