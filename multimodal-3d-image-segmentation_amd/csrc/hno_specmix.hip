// Shared-weight spectral channel mix: L stacked layers  z <- act((W_l [+ I]) z)  over the kept
// modes, as ONE forward kernel and ONE backward kernel (+ the slab reduce).
//
// Reference: NeuralOperatorBlock.forward x n_XS (nets/hnosegxs.py:307-329) around
// HartleyOperator._call3d_notransform (nets/hartley_operator.py:287-292): per layer an
// einsum 'oi,bidhw->bodhw', an add and a SELU over a (B, C, 2m0, 2m1, 2m2) block -- 1.5 MB per
// sample at the benchmark size, so the reference (and a per-layer kernel chain) is purely
// launch/latency-bound there.  The layers are pointwise over modes, hence the whole stack runs
// on one 32-mode tile held in registers:
//
//   v_mfma_f32_32x32x2_f32 with the weights as the A operand and the tile as the B operand.
//   The k-slot (ks, h = lane >> 5) of the B operand is assigned to channel
//        chan(ks, h) = (ks & 3) + 8 * (ks >> 2) + 4 * h,
//   which is exactly the row that accumulator register r = ks of lane-half h holds.  The output
//   of one layer therefore IS the B operand of the next one, register for register: no LDS, no
//   cross-lane traffic between layers.  Global accesses are two 128-byte row segments per
//   instruction.
//
// Backward (per tile, layers in reverse): g <- g * act'(z_l);  dW_l += g z_{l-1}^T (contraction
// over modes: both operands go through a wave-private LDS tile and 16x16x4 MFMAs);
// g <- (W_l [+ I])^T g (32x32x2 MFMA, same register identity as the forward).  Weight-gradient
// partials: one slab per block, summed in fixed order by reduce_partials_kernel (no atomics).
#include "hno_common.h"

namespace hno {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mix_mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

#define MIX_MAX_L 4   // layers per launch (register-resident weights); longer stacks are chunked
#define MIX_LD 34     // LDS row stride: == 2 (mod 4) -> conflict-free 16x16x4 operand reads

struct MixArgs {
    const float *zin;             // input of the first layer of this launch (B, C, M)
    const float *W[MIX_MAX_L];    // (C, C) each
    float *zs;                    // outputs of the layers of this launch, each (B, C, M), stacked
    const float *g;               // backward: dL/d(output of the last layer of this launch)
    float *gz;                    // backward: dL/d(zin)
    float *partials;              // backward: gridDim.x slabs of L*C*C floats
    int B, C, L, residual, act;
    unsigned M;
};

__device__ __forceinline__ int mix_chan(int ks, int h) { return (ks & 3) + 8 * (ks >> 2) + 4 * h; }

// one wave per block; tile = 32 modes of one sample
template <int NK, int LT>
__global__ __launch_bounds__(64) void specmix_fwd_kernel(MixArgs a) {
    const int lane = threadIdx.x, h = lane >> 5, c = lane & 31;
    const int C = a.C;
    const unsigned M = a.M;
    float w[LT][NK];   // A operand: W'[row = c][k-slot -> chan(ks, h)]
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int ch = mix_chan(ks, h);
            const bool ok = c < C && ch < C;
            w[l][ks] = ok ? a.W[l][(size_t)c * C + ch] : 0.f;
            if (a.residual && ok && c == ch) w[l][ks] += 1.f;
        }
    const unsigned tiles_per_b = (M + 31) / 32, ntiles = tiles_per_b * a.B;
    const size_t sample = (size_t)C * M, layer = sample * a.B;
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const unsigned b = t / tiles_per_b;
        const unsigned col = (t - b * tiles_per_b) * 32 + c;
        const bool cin = col < M;
        const float *zb = a.zin + b * sample;
        float z[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int ch = mix_chan(ks, h);
            z[ks] = (ch < C && cin) ? zb[(size_t)ch * M + col] : 0.f;
        }
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc = mix_mfma32(w[l][ks], z[ks], acc);
            float *ob = a.zs + l * layer + b * sample;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int ch = mix_chan(ks, h);
                z[ks] = act_apply(acc[ks], a.act);   // rows >= C have zero weights: act(0) = 0
                if (ch < C && cin) ob[(size_t)ch * M + col] = z[ks];
            }
        }
    }
}

// Compact form of the forward kernel: the layer loop is NOT unrolled and the weights of layer l + 1 are
// fetched while layer l computes.  Each wave runs its code exactly once (one tile per wave at the
// benchmark size), so a fully unrolled 16 KB instruction stream is paid for in instruction-cache
// misses; this one is a ~2 KB loop body.
template <int NK, bool EXACT>   // EXACT: C == 2 * NK and M % 32 == 0 (no guards)
__global__ __launch_bounds__(64) void specmix_fwd_loop_kernel(MixArgs a) {
    const int lane = threadIdx.x, h = lane >> 5, c = lane & 31;
    const int C = EXACT ? 2 * NK : a.C;
    const unsigned M = a.M;
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const float res = a.residual ? 1.f : 0.f;
    // lane part of the weight offset.  Lanes c >= C (EXACT skips the guards) feed output rows that are never stored: they re-read row C - 1
    // instead of running up to (32 - C) rows past the end of W (a fault when W ends its allocator segment)
    const unsigned wl = (unsigned)(c < C ? c : C - 1) * C + 4 * h;
    const int dsel = c - 4 * h;                         // diagonal: c == row(ks) + 4h
    const unsigned tiles_per_b = (M + 31) / 32, ntiles = tiles_per_b * a.B;
    const size_t sample = (size_t)C * M, layer = sample * a.B;
    auto load_w = [&](const float *Wp, float (&w)[NK]) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int row = (ks & 3) + 8 * (ks >> 2);
            const bool ok = EXACT || (c < C && row + 4 * h < C);
            w[ks] = ok ? Wp[wl + row] : 0.f;
            if (ok) w[ks] += (dsel == row) ? res : 0.f;
        }
    };
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const unsigned b = t / tiles_per_b;
        const unsigned col = (t - b * tiles_per_b) * 32 + c;
        const bool cin = EXACT || col < M;
        const unsigned zoff = 4 * h * M + (cin ? col : 0u);
        const float *zb = a.zin + b * sample;
        float w[NK], z[NK];
        load_w(a.W[0], w);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int row = (ks & 3) + 8 * (ks >> 2);
            const bool ok = EXACT || (row + 4 * h < C && cin);
            z[ks] = ok ? (zb + (size_t)row * M)[zoff] : 0.f;
        }
        float *ob = a.zs + b * sample;
#pragma unroll 1
        for (int l = 0; l < a.L; ++l) {
            float wn[NK];
            if (l + 1 < a.L) load_w(l == 0 ? a.W[1] : (l == 1 ? a.W[2] : a.W[3]), wn);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc = mix_mfma32(w[ks], z[ks], acc);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int row = (ks & 3) + 8 * (ks >> 2);
                const float x = acc[ks];
                z[ks] = (x > 0.f || lin) ? ap * x : aq * neg_expm1(x);
                if (EXACT || (row + 4 * h < C && cin)) (ob + (size_t)row * M)[zoff] = z[ks];
            }
            ob += layer;
            if (l + 1 < a.L) {
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) w[ks] = wn[ks];
            }
        }
    }
}

template <int NK, int LT>
__global__ __launch_bounds__(256) void specmix_bwd_kernel(MixArgs a) {
    extern __shared__ float lds[];
    constexpr int CT = 2 * NK, MT = (CT + 15) / 16;   // padded channel count, 16-row tiles
    constexpr int TILE = 2 * MT * 16 * MIX_LD;        // G and Z tiles of one wave
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int C = a.C;
    const unsigned M = a.M;
    float *G = lds + (size_t)wave * TILE;   // [o][mode]
    float *Z = G + MT * 16 * MIX_LD;        // [i][mode]
    float wt[LT][NK];   // A operand of the input gradient: W'^T[row = i = c][k-slot -> o = chan(ks, h)]
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int o = mix_chan(ks, h);
            const bool ok = c < C && o < C;
            wt[l][ks] = ok ? a.W[l][(size_t)o * C + c] : 0.f;
            if (a.residual && ok && c == o) wt[l][ks] += 1.f;
        }
    for (int i = lane; i < TILE; i += 64) G[i] = 0.f;
    f32x4 dw[LT][MT][MT];
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < MT; ++n) dw[l][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned tiles_per_b = (M + 31) / 32, ntiles = tiles_per_b * a.B;
    const size_t sample = (size_t)C * M, layer = sample * a.B;
    const float *ga = G + (lane & 15) * MIX_LD + (lane >> 4);
    const float *za = Z + (lane & 15) * MIX_LD + (lane >> 4);
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
        const unsigned b = t / tiles_per_b;
        const unsigned col = (t - b * tiles_per_b) * 32 + c;
        const bool cin = col < M;
        float g[NK], zt[LT + 1][NK];   // zt[0] = input of the first layer, zt[l + 1] = output of layer l
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int ch = mix_chan(ks, h);
            const bool ok = ch < C && cin;
            const size_t off = b * sample + (size_t)ch * M + col;
            g[ks] = ok ? a.g[off] : 0.f;
            zt[0][ks] = ok ? a.zin[off] : 0.f;
#pragma unroll
            for (int l = 0; l < LT; ++l) zt[l + 1][ks] = ok ? a.zs[l * layer + off] : 0.f;
        }
#pragma unroll
        for (int l = LT - 1; l >= 0; --l) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int ch = mix_chan(ks, h);
                g[ks] *= act_grad_from_out(zt[l + 1][ks], a.act);
                if (ch < MT * 16) {
                    G[ch * MIX_LD + c] = g[ks];
                    Z[ch * MIX_LD + c] = zt[l][ks];
                }
            }
            // wave-private tile: LDS operations of one wave execute in order, only the compiler has to
            // be kept from moving the reads above the writes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 2
            for (int kk = 0; kk < 8; ++kk) {
                float av[MT], bv[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = ga[m * 16 * MIX_LD + kk * 4];
#pragma unroll
                for (int n = 0; n < MT; ++n) bv[n] = za[n * 16 * MIX_LD + kk * 4];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < MT; ++n) dw[l][m][n] = mfma16(av[m], bv[n], dw[l][m][n]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc = mix_mfma32(wt[l][ks], g[ks], acc);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) g[ks] = acc[ks];
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int ch = mix_chan(ks, h);
            if (ch < C && cin) a.gz[b * sample + (size_t)ch * M + col] = g[ks];
        }
    }
    // flush: per-wave fragments -> LDS -> one slab per block
    const int n = LT * C * C;
    __syncthreads();   // the scratch area overlaps the other waves' tiles
    float *mine = lds + (size_t)wave * n;
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < MT; ++nn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = m * 16 + (lane >> 4) * 4 + r, i = nn * 16 + (lane & 15);
                    if (o < C && i < C) mine[(l * C + o) * C + i] = dw[l][m][nn][r];
                }
    block_sum_to_slab(lds, n, a.partials + (size_t)blockIdx.x * n, threadIdx.x, 4);
}

// Compact backward: runtime layer loop (one layer ahead prefetch of the saved activations and the
// weights), per-layer weight-gradient fragments accumulated into a wave-private LDS scratch instead
// of L register sets.  Same reasoning as specmix_fwd_loop_kernel: each wave runs once.
template <int NK, bool EXACT>
__global__ __launch_bounds__(256) void specmix_bwd_loop_kernel(MixArgs a) {
    extern __shared__ float lds[];
    constexpr int CT = 2 * NK, MT = (CT + 15) / 16;
    constexpr int TILE = 2 * MT * 16 * MIX_LD;
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int C = EXACT ? CT : a.C;
    const unsigned M = a.M;
    const int L = a.L, n = L * C * C;
    float *G = lds + (size_t)wave * TILE;   // [o][mode]
    float *Z = G + MT * 16 * MIX_LD;        // [i][mode]
    float *mine = lds + 4 * TILE + (size_t)wave * n;
    for (int i = lane; i < TILE; i += 64) G[i] = 0.f;
    for (int i = lane; i < n; i += 64) mine[i] = 0.f;
    const float dp = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float dq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const float res = a.residual ? 1.f : 0.f;
    const unsigned wlt = 4u * h * C + (c < C ? c : C - 1);   // lane part of W^T: W[(row + 4h) * C + c] (lanes c >= C: unused, kept inside W)
    const int dsel = c - 4 * h;
    const unsigned tiles_per_b = (M + 31) / 32, ntiles = tiles_per_b * a.B;
    const size_t sample = (size_t)C * M, layer = sample * a.B;
    const float *ga = G + (lane & 15) * MIX_LD + (lane >> 4);
    const float *za = Z + (lane & 15) * MIX_LD + (lane >> 4);
    auto load_w = [&](const float *Wp, float (&w)[NK]) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int row = (ks & 3) + 8 * (ks >> 2);
            const bool ok = EXACT || (c < C && row + 4 * h < C);
            w[ks] = ok ? (Wp + (size_t)row * C)[wlt] : 0.f;
            if (ok) w[ks] += (dsel == row) ? res : 0.f;
        }
    };
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
        const unsigned b = t / tiles_per_b;
        const unsigned col = (t - b * tiles_per_b) * 32 + c;
        const bool cin = EXACT || col < M;
        const unsigned zoff = 4 * h * M + (cin ? col : 0u);
        auto load_z = [&](const float *base, float (&z)[NK]) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int row = (ks & 3) + 8 * (ks >> 2);
                const bool ok = EXACT || (row + 4 * h < C && cin);
                z[ks] = ok ? (base + (size_t)row * M)[zoff] : 0.f;
            }
        };
        float g[NK], zo[NK], zi[NK], wt[NK];
        load_z(a.g + b * sample, g);
        load_z(a.zs + (size_t)(L - 1) * layer + b * sample, zo);
        load_z((L > 1 ? a.zs + (size_t)(L - 2) * layer : a.zin) + b * sample, zi);
        load_w(L == 1 ? a.W[0] : (L == 2 ? a.W[1] : (L == 3 ? a.W[2] : a.W[3])), wt);
#pragma unroll 1
        for (int l = L - 1; l >= 0; --l) {
            float zn[NK], wn[NK];
            if (l > 0) {
                load_z((l > 1 ? a.zs + (size_t)(l - 2) * layer : a.zin) + b * sample, zn);
                load_w(l == 1 ? a.W[0] : (l == 2 ? a.W[1] : a.W[2]), wn);
            }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int ch = (ks & 3) + 8 * (ks >> 2) + 4 * h;
                g[ks] *= (zo[ks] > 0.f || lin) ? dp : zo[ks] + dq;
                G[ch * MIX_LD + c] = g[ks];
                Z[ch * MIX_LD + c] = zi[ks];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x4 dw[MT][MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < MT; ++nn) dw[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float av[MT], bv[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = ga[m * 16 * MIX_LD + kk * 4];
#pragma unroll
                for (int nn = 0; nn < MT; ++nn) bv[nn] = za[nn * 16 * MIX_LD + kk * 4];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nn = 0; nn < MT; ++nn) dw[m][nn] = mfma16(av[m], bv[nn], dw[m][nn]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float *dst = mine + (size_t)l * C * C;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < MT; ++nn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = m * 16 + (lane >> 4) * 4 + r, i = nn * 16 + (lane & 15);
                        if (o < C && i < C) dst[o * C + i] += dw[m][nn][r];
                    }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc = mix_mfma32(wt[ks], g[ks], acc);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                g[ks] = acc[ks];
                zo[ks] = zi[ks];
            }
            if (l > 0) {
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    zi[ks] = zn[ks];
                    wt[ks] = wn[ks];
                }
            }
        }
        float *gzb = a.gz + b * sample;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int row = (ks & 3) + 8 * (ks >> 2);
            if (EXACT || (row + 4 * h < C && cin)) (gzb + (size_t)row * M)[zoff] = g[ks];
        }
    }
    block_sum_to_slab(lds + 4 * TILE, n, a.partials + (size_t)blockIdx.x * n, threadIdx.x, 4);
}

template <int NK>
static void mix_bwd_loop_dispatch(bool exact, int grid, size_t lds, hipStream_t s, const MixArgs &a) {
    static int attr_done = -1;
    if (attr_done != current_device()) {
        (void)hipFuncSetAttribute((const void *)specmix_bwd_loop_kernel<NK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)specmix_bwd_loop_kernel<NK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = current_device();
    }
    if (exact) hipLaunchKernelGGL((specmix_bwd_loop_kernel<NK, true>), dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((specmix_bwd_loop_kernel<NK, false>), dim3(grid), dim3(256), lds, s, a);
}

template <int NK>
static void mix_fwd_dispatch(int LT, int grid, hipStream_t s, const MixArgs &a) {
    switch (LT) {
        case 1: hipLaunchKernelGGL((specmix_fwd_kernel<NK, 1>), dim3(grid), dim3(64), 0, s, a); break;
        case 2: hipLaunchKernelGGL((specmix_fwd_kernel<NK, 2>), dim3(grid), dim3(64), 0, s, a); break;
        case 3: hipLaunchKernelGGL((specmix_fwd_kernel<NK, 3>), dim3(grid), dim3(64), 0, s, a); break;
        default: hipLaunchKernelGGL((specmix_fwd_kernel<NK, 4>), dim3(grid), dim3(64), 0, s, a); break;
    }
}

template <int NK>
static void mix_bwd_dispatch(int LT, int grid, size_t lds, hipStream_t s, const MixArgs &a) {
    static int attr_done = -1;
    if (attr_done != current_device()) {
        (void)hipFuncSetAttribute((const void *)specmix_bwd_kernel<NK, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)specmix_bwd_kernel<NK, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)specmix_bwd_kernel<NK, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)specmix_bwd_kernel<NK, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = current_device();
    }
    switch (LT) {
        case 1: hipLaunchKernelGGL((specmix_bwd_kernel<NK, 1>), dim3(grid), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL((specmix_bwd_kernel<NK, 2>), dim3(grid), dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL((specmix_bwd_kernel<NK, 3>), dim3(grid), dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL((specmix_bwd_kernel<NK, 4>), dim3(grid), dim3(256), lds, s, a); break;
    }
}

// implemented in hno_pwconv.hip (per-layer path for C > 32)
int pwconv_fwd_launch(const float *xa, int Ca, const float *xb, int Cb, const float *W, const float *bias, float *y, int B,
                      int Cout, long long V, int act, int residual, void *stream);
int pwconv_bwd_launch(const float *gy, const float *y, const float *xa, int Ca, const float *xb, int Cb, const float *W,
                      float *gxa, float *gxb, float *dW, float *dbias, void *workspace, int B, int Cout, long long V, int act,
                      int residual, void *stream, int xa_act, int accumulate_gx, const float *Wbr = nullptr);

static int specmix_fwd(const float *z0, const float *const *Wl, float *zs, int B, int C, int M, int L, int residual, int act,
                       void *stream) {
    const size_t layer = (size_t)B * C * M;
    if (C > 32) {
        for (int l = 0; l < L; ++l) {
            const float *in = l == 0 ? z0 : zs + (size_t)(l - 1) * layer;
            int rc = pwconv_fwd_launch(in, C, nullptr, 0, Wl[l], nullptr, zs + (size_t)l * layer, B, C, M, act, residual, stream);
            if (rc) return rc;
        }
        return HNO_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    const long long ntiles = (long long)((M + 31) / 32) * B;
    const int grid = (int)(ntiles < 4096 ? ntiles : 4096);
    for (int l0 = 0; l0 < L; l0 += MIX_MAX_L) {
        const int cnt = L - l0 < MIX_MAX_L ? L - l0 : MIX_MAX_L;
        MixArgs a = {};
        a.zin = l0 == 0 ? z0 : zs + (size_t)(l0 - 1) * layer;
        for (int j = 0; j < cnt; ++j) a.W[j] = Wl[l0 + j];
        a.zs = zs + (size_t)l0 * layer;
        a.B = B; a.C = C; a.L = cnt; a.residual = residual; a.act = act; a.M = (unsigned)M;
        ProfScope ps(KID_SPECMIX_FWD, s, 4.0 * (double)layer * (1 + cnt));
        if (!(debug_flags() & 16)) {
            const bool exact = M % 32 == 0;
            if (C <= 16) {
                if (exact && C == 16) hipLaunchKernelGGL((specmix_fwd_loop_kernel<8, true>), dim3(grid), dim3(64), 0, s, a);
                else hipLaunchKernelGGL((specmix_fwd_loop_kernel<8, false>), dim3(grid), dim3(64), 0, s, a);
            } else if (C <= 24) {
                if (exact && C == 24) hipLaunchKernelGGL((specmix_fwd_loop_kernel<12, true>), dim3(grid), dim3(64), 0, s, a);
                else hipLaunchKernelGGL((specmix_fwd_loop_kernel<12, false>), dim3(grid), dim3(64), 0, s, a);
            } else {
                if (exact && C == 32) hipLaunchKernelGGL((specmix_fwd_loop_kernel<16, true>), dim3(grid), dim3(64), 0, s, a);
                else hipLaunchKernelGGL((specmix_fwd_loop_kernel<16, false>), dim3(grid), dim3(64), 0, s, a);
            }
        } else if (C <= 16) mix_fwd_dispatch<8>(cnt, grid, s, a);
        else if (C <= 24) mix_fwd_dispatch<12>(cnt, grid, s, a);
        else mix_fwd_dispatch<16>(cnt, grid, s, a);
        HNO_CHECK_LAUNCH();
    }
    return HNO_OK;
}

static int specmix_bwd(const float *g, const float *z0, const float *zs, const float *const *Wl, float *gz0, float *dW,
                       void *workspace, int B, int C, int M, int L, int residual, int act, void *stream) {
    const size_t layer = (size_t)B * C * M;
    if (C > 32) {
        // wide stacks run layer by layer; the chunked pointwise backward re-reads its output gradient
        // once per block of 32 output channels, so the gradient ping-pongs between gz0 and a temporary
        // placed behind the slabs in the workspace (layer 0 lands in gz0)
        float *tmp = (float *)((char *)workspace + hno_pwconv_bwd_workspace_bytes(C, C));
        const int was_deferring = hno_set_defer_reduce(0);   // the layers share one slab workspace: reduce each at once
        struct Restore { int v; ~Restore() { hno_set_defer_reduce(v); } } restore{was_deferring};
        for (int l = L - 1; l >= 0; --l) {
            const float *in = l == 0 ? z0 : zs + (size_t)(l - 1) * layer;
            const float *gy = l == L - 1 ? g : ((l + 1) % 2 == 0 ? gz0 : tmp);
            float *gx = l % 2 == 0 ? gz0 : tmp;
            int rc = pwconv_bwd_launch(gy, zs + (size_t)l * layer, in, C, nullptr, 0, Wl[l], gx, nullptr,
                                       dW + (size_t)l * C * C, nullptr, workspace, B, C, M, act, residual, stream, HNO_ACT_NONE, 0);
            if (rc) return rc;
        }
        return HNO_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    const long long ntiles = (long long)((M + 31) / 32) * B;
    int grid = (int)((ntiles + 3) / 4);
    if (grid > 256) grid = 256;   // grid * MIX_MAX_L slabs of C*C floats fit hno_pwconv_bwd_workspace_bytes(C, C)
    const int nk = C <= 16 ? 8 : (C <= 24 ? 12 : 16);
    const int mt = (2 * nk + 15) / 16;
    const int nchunks = (L + MIX_MAX_L - 1) / MIX_MAX_L;
    for (int ci = nchunks - 1; ci >= 0; --ci) {
        const int l0 = ci * MIX_MAX_L;
        const int cnt = L - l0 < MIX_MAX_L ? L - l0 : MIX_MAX_L;
        MixArgs a = {};
        a.zin = l0 == 0 ? z0 : zs + (size_t)(l0 - 1) * layer;
        for (int j = 0; j < cnt; ++j) a.W[j] = Wl[l0 + j];
        a.zs = const_cast<float *>(zs) + (size_t)l0 * layer;
        a.g = ci == nchunks - 1 ? g : gz0;   // in place: a wave reads its tile before it writes it
        a.gz = gz0;
        a.partials = (float *)workspace;
        a.B = B; a.C = C; a.L = cnt; a.residual = residual; a.act = act; a.M = (unsigned)M;
        const int n = cnt * C * C;
        {
            ProfScope ps(KID_SPECMIX_BWD, s, 4.0 * (double)layer * (cnt + 3));
            if (!(debug_flags() & 16)) {
                const size_t lds = sizeof(float) * 4 * (size_t)(2 * mt * 16 * MIX_LD + n);
                const bool exact = M % 32 == 0 && C == 2 * nk;
                if (nk == 8) mix_bwd_loop_dispatch<8>(exact, grid, lds, s, a);
                else if (nk == 12) mix_bwd_loop_dispatch<12>(exact, grid, lds, s, a);
                else mix_bwd_loop_dispatch<16>(exact, grid, lds, s, a);
            } else {
                size_t lds = sizeof(float) * 4 * (size_t)(2 * mt * 16 * MIX_LD);
                if (lds < sizeof(float) * 4 * (size_t)n) lds = sizeof(float) * 4 * (size_t)n;
                if (nk == 8) mix_bwd_dispatch<8>(cnt, grid, lds, s, a);
                else if (nk == 12) mix_bwd_dispatch<12>(cnt, grid, lds, s, a);
                else mix_bwd_dispatch<16>(cnt, grid, lds, s, a);
            }
        }
        HNO_CHECK_LAUNCH();
        int rc = reduce_partials_launch(a.partials, grid, n, dW + (size_t)l0 * C * C, n, nullptr, s, 0, 0, L <= MIX_MAX_L);   // one round only
        if (rc) return rc;
    }
    return HNO_OK;
}

}  // namespace hno

using namespace hno;

#define MIX_MAX_LAYERS 64

extern "C" size_t hno_specmix_bwd_workspace_bytes(int B, int C, int M, int L) {
    size_t n = hno_pwconv_bwd_workspace_bytes(C, C);
    if (C > 32 && L > 1) n += sizeof(float) * (size_t)B * C * M;
    return n;
}

extern "C" int hno_specmix_layers_fwd(const float *z0, const float *const *W_layers, float *zs, int B, int C, int M, int L,
                                      int residual, int act, void *stream) {
    HNO_REQUIRE(z0 && W_layers && zs && B > 0 && C > 0 && M > 0 && L > 0, "hno_specmix_layers_fwd: bad argument");
    for (int l = 0; l < L; ++l) HNO_REQUIRE(W_layers[l], "hno_specmix_layers_fwd: W_layers[%d] is NULL", l);
    return specmix_fwd(z0, W_layers, zs, B, C, M, L, residual, act, stream);
}

extern "C" int hno_specmix_layers_bwd(const float *g, const float *z0, const float *zs, const float *const *W_layers,
                                      float *gz0, float *dW, void *workspace, int B, int C, int M, int L, int residual,
                                      int act, void *stream) {
    HNO_REQUIRE(g && z0 && zs && W_layers && gz0 && dW && workspace && B > 0 && C > 0 && M > 0 && L > 0,
                "hno_specmix_layers_bwd: bad argument");
    for (int l = 0; l < L; ++l) HNO_REQUIRE(W_layers[l], "hno_specmix_layers_bwd: W_layers[%d] is NULL", l);
    // bit 8 of residual: record the slab reduction for hno_flush_reduces
    const int prev = hno_set_defer_reduce(0);
    struct Restore { int v; ~Restore() { hno_set_defer_reduce(v); } } restore{prev};
    hno_set_defer_reduce(((residual >> 8) & 1) ? 1 : prev);
    return specmix_bwd(g, z0, zs, W_layers, gz0, dW, workspace, B, C, M, L, residual & 0xff, act, stream);
}

// Stacked-weight form: W is (L, C, C) contiguous.
extern "C" int hno_specmix_shared_fwd(const float *z0, const float *W, float *zs, int B, int C, int M, int L, int residual,
                                      int act, void *stream) {
    HNO_REQUIRE(z0 && W && zs && B > 0 && C > 0 && M > 0 && L > 0 && L <= MIX_MAX_LAYERS, "hno_specmix_shared_fwd: bad argument");
    const float *Wl[MIX_MAX_LAYERS];
    for (int l = 0; l < L; ++l) Wl[l] = W + (size_t)l * C * C;
    return specmix_fwd(z0, Wl, zs, B, C, M, L, residual, act, stream);
}

extern "C" int hno_specmix_shared_bwd(const float *g, const float *z0, const float *zs, const float *W, float *gz0, float *dW,
                                      void *workspace, int B, int C, int M, int L, int residual, int act, void *stream) {
    HNO_REQUIRE(g && z0 && zs && W && gz0 && dW && workspace && B > 0 && C > 0 && M > 0 && L > 0 && L <= MIX_MAX_LAYERS,
                "hno_specmix_shared_bwd: bad argument");
    const float *Wl[MIX_MAX_LAYERS];
    for (int l = 0; l < L; ++l) Wl[l] = W + (size_t)l * C * C;
    return specmix_bwd(g, z0, zs, Wl, gz0, dW, workspace, B, C, M, L, residual, act, stream);
}
