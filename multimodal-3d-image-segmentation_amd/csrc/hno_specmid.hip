// The spectral middle of an HNO-XS block in ONE kernel per direction:
//
//   forward :  axis-D transform + Re -/+ Im + crop  ->  L x ( z <- act((W + I) z) )  ->  pad + axis-D inverse transform
//   backward:  the same two D steps (PadInverse^T = TransformCrop and vice versa) around the BACKWARD of the layer stack: per layer
//              g <- g * act'(z_l), dW_l += g z_{l-1}^T, g <- (W_l + I)^T g; the gradients of the cropped spectra never reach memory
//
// i.e. what dht_fwd_d_kernel, specmix_fwd_loop_kernel and dht_inv_d_kernel (hno_dht.hip, hno_specmix.hip) do in three launches
// between the two plane transforms.  Reference: TransformCrop.forward (nets/hnosegxs.py:378-410, axis D of dhtn, nets/dht.py:16-36),
// NeuralOperatorBlock.forward x n_XS (nets/hnosegxs.py:307-329, HartleyOperator._call3d_notransform, nets/hartley_operator.py:287-292),
// PadInverse.forward (nets/hnosegxs.py:454-494).
//
// Why: each of the three kernels is one memory round trip plus a launch (10 + 12 + 10 us for ~1 us of arithmetic each, 16 chains per
// HNOSeg-XS step = 17 % of the step, profiles/r02_d_*).  The chain is closed per COLUMN of the intermediate: column (k1, k2 >= 0) of
// the plane-transform output, all N0 planes and all channels of one sample, determines the 40 kept modes (+-k0, k1, k2) and
// (+-k0, -k1, -k2), the channel mix is pointwise over modes, and the inverse D step of the same column reads exactly those 40 modes.
// A workgroup owns 4 adjacent k2 columns of one (sample, k1): 2 x 29 x 4 = 232 workgroups at the benchmark size.
//   * the D steps run on the VALU: 65 -> 11 cos / 10 sin sums per (channel, part, column) is 683 FMAs, far too little per column for
//     a 16-column MFMA tile (round 2's fused attempt kept the MFMA tiles and ended with 58 workgroups, DESIGN lesson 29); the twiddles
//     are wave-uniform, i.e. scalar operands;
//   * the channel mix is the register-chained v_mfma_f32_32x32x2_f32 stack of hno_specmix.hip on the 160 modes of the tile;
//   * the workgroup reads its columns of the plane-transform output and overwrites the same addresses with the inverse D step's
//     output: the workspace is transformed in place, no second buffer.
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "hno_common.h"

namespace hno {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MidArgs {
    float *ws;                 // [b * C + c][n0][part 2][k1s (2 m1 + 1)][16]: in = forward plane transform, out = inverse D step
    const float *W[4];         // (C, C) per layer
    float *zs;                 // (L + 1) stacked (B, C, 2 m0, 2 m1, 2 m2): z0 = cropped spectrum, z_l = output of layer l
    const float *tw;           // twiddle A operands of the two D steps in lane order (mid_twiddles)
    int B, L, residual, act;
    int m1, m2;
    float scale;
    float *partials;           // backward: one slab of L x C x C weight-gradient partial sums per workgroup
    long long *stamps;         // phase stamps of workgroup 0, wave 0 (debug flag 1024)
    int dbg;                   // timing aids (results wrong): 1 = no forward D arithmetic, 2 = no layers, 4 = no inverse D step, 8 = no loads
    int zl;                    // layout of ws (hno_dht.hip, DhtArgs.zl): 1 = [k1 position][k2 tile][plane][part][4]
};

__device__ __forceinline__ int mid_chan(int ks, int h) { return (ks & 3) + 8 * (ks >> 2) + 4 * h; }

// N0: planes (odd, or even since round 5), M0: kept modes along D (frequencies k0 = 0..M0 are computed; +M0 itself is not kept, -M0 is)
//
// Both D steps are small GEMMs on v_mfma_f32_16x16x4_f32 whose 16 COLUMNS are (column j of the tile, re / im part, 2 channels): the
// N dimension of the MFMA is filled by batching over parts and channels, not by widening the k2 tile, so 4-column tiles cost no
// matrix-core work.  Twiddles are the A operands (registers, loaded once in lane order).
//   forward : rows k0 = 0..15, K = n (folded: f[n] = v[n] + v[N0 - n] against cos, d[n] = v[n] - v[N0 - n] against sin), 9 k-steps
//   inverse : rows n = 1 + 16 mt + i, K = k0 (3 k-steps); n = 0 is the plain sum of the cosine coefficients
// waves per workgroup of the Hartley kernel.  Round 3 ran 8: the 12 column tiles of the two D steps then took two passes on four of the
// waves (phase 1 ended when the two-tile waves did).  12 waves = one tile each, three waves per SIMD; one workgroup per CU either way
// (232 workgroups at the benchmark size).
#define MID_NW 12
template <int N0, int M0, bool BWD, bool ZLAY>   // ZLAY: layout of the intermediate (MidArgs.zl) at compile time: plane stride 8 floats -> immediate offsets
__global__ __launch_bounds__(64 * MID_NW, 1) void spec_mid_kernel(MidArgs a) {
    constexpr int NW = MID_NW, NTH = 64 * NW;          // waves per workgroup: one D-step column tile per wave (NCT = 12)
    constexpr int C = 24, NK = 12, J = N0 / 2, K0 = M0 + 1;
    constexpr int KC = (J + 1 + 3) / 4;               // k-steps of the forward D step (n = 0 .. 4 KC - 1)
    constexpr int KI = (K0 + 3) / 4;                   // k-steps of the inverse D step (k0 = 0 .. 4 KI - 1)
    constexpr int NMT = (J + 15) / 16;                 // 16-row output tiles of the inverse D step (n = 1 .. 16 NMT; rows n > J masked)
    constexpr int NMODE = 2 * 2 * M0 * 4;              // modes of the tile: (sign of (k1, k2), o0, column)
    constexpr int NT = (NMODE + 31) / 32;             // 32-mode MFMA tiles
    constexpr int NCT = C / 2;                         // column tiles of the D steps: 2 channels x 2 parts x 4 columns
    static_assert(K0 <= 16 && 2 * M0 <= N0, "one k0 tile");      // (an even N0: the plane N0 / 2 is its own mirror -- see mid_twiddles)
    constexpr int NPQ = C * 2 * 2 * K0 * 4, NZL = C * NT * 32;
    constexpr int GLD = 34;                            // row stride of the G / Z tiles: == 2 (mod 4), conflict-free 16x16x4 operand reads
    constexpr int TILE = 2 * 32 * GLD;                 // backward: [o][mode] and [i][mode] tiles of one wave
    constexpr int WLD = 25, NWL = 4 * 32 * WLD;        // layer weights in LDS: [l][row 32][25] (odd pitch: conflict-free both ways)
    extern __shared__ float lds[];
    float *WL = lds;                                   // W'_l = W_l + residual identity, rows >= C zero
    float *PQ = lds + NWL;                             // [c][part][P | Q][k0][column]
    float *ZL = PQ + NPQ;                              // [c][mode]: output of the last layer (backward: gradient of the first layer's input)
    float *GZ = ZL + NZL;                              // backward: NT tiles
    float *MINE = GZ + NT * TILE;                      // backward: per-wave weight-gradient fragments [wave][l][o][i]
    if (a.dbg & 16) return;
    HNO_STAMP(a.stamps, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    // workgroup -> (sample, k1 row, k2 tile).  The 8 workgroups that share the 128-byte lines of a k1 row pair get ids that are
    // equal mod 8, i.e. the same XCD under round-robin dispatch: each line then crosses the fabric once instead of eight times
    // (speed only; nothing depends on the placement)
    const int m1 = a.m1, m2 = a.m2, K1S = 2 * m1 + 1, CP = K1S * 16;
    const int pairs = (K1S + 1) / 2, G8 = (a.B * pairs + 7) & ~7;
    const int w8 = blockIdx.x / G8, gidx = blockIdx.x - w8 * G8;
    const int b = gidx / pairs, k1s = 2 * (gidx - b * pairs) + (w8 >> 2), kt2 = w8 & 3;
    if (b >= a.B || k1s >= K1S) {
        if (BWD)
            for (int i = tid; i < a.L * C * C; i += NTH) a.partials[(size_t)blockIdx.x * (a.L * C * C) + i] = 0.f;
        return;
    }
    const int k1 = k1s - m1;
    // intermediate layout (hno_dht.hip, DhtArgs.zl): 0 = [plane][part][k1 position][16], 1 = [k1 position][k2 tile][plane][part][4]
    const size_t pstride = ZLAY ? (size_t)8 : (size_t)2 * CP;            // floats per n0 plane
    // float offset of (channel c, plane 0, part, column j) of this workgroup's columns
    auto col_base = [&](int c, int part_, int j_) -> size_t {
        if constexpr (ZLAY) return ((size_t)(k1s * 4 + kt2) * ((size_t)a.B * C * N0) + (size_t)(b * C + c) * N0) * 8 + part_ * 4 + j_;
        return ((size_t)(b * C + c) * N0) * ((size_t)2 * CP) + (size_t)part_ * CP + k1s * 16 + kt2 * 4 + j_;
    };
    // column of the D-step tiles held by this lane: j = column of the k2 tile, re / im part, channel within the pair
    const int j = l15 & 3, part = (l15 >> 2) & 1, cloc = l15 >> 3;
    // twiddle A operands in lane order: forward cos / sin [ks], inverse cos / sin [mt][ks]
    float tcF[KC], tsF[KC], tcI[NMT][KI], tsI[NMT][KI];
    {
        const float *tb = a.tw + lane;
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
            tcF[ks] = tb[ks * 64];
            tsF[ks] = tb[(KC + ks) * 64];
        }
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KI; ++ks) {
                tcI[mt][ks] = tb[(2 * KC + (mt * 2 + 0) * KI + ks) * 64];
                tsI[mt][ks] = tb[(2 * KC + (mt * 2 + 1) * KI + ks) * 64];
            }
    }
    // ---------------- layer weights as the A operand of the 32x32x2 MFMA: lane (row cl, k-slot half h) holds W'[cl][chan(ks, h)]
    // (backward: the transpose, W'[chan(ks, h)][cl]).  Round 4: all layers' weights go to LDS once, at the top, by every wave together,
    // as W' = W + residual identity with rows >= C zero.  Round 3 fetched a layer's weights from global memory one layer ahead -- and
    // the compiler's wait for them (vmcnt counts loads AND stores, in order) also waited for the twelve scattered stores of the layer
    // output issued in between: ~1 600 cycles of exposed store latency per layer (in-kernel stamps; 3 of the forward's 21 us).  LDS
    // reads are counted separately (lgkmcnt), so the stores now leave and nobody waits for them.
    const int h = lane >> 5, cl = lane & 31;
    // forward: W'[cl][chan(ks, h)];  backward: W'[chan(ks, h)][cl] (cl >= C: rows of zeros above -> read column cl of a zero... see below)
    auto read_w = [&](int l, float (&w)[NK]) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int ch = mid_chan(ks, h);
            if (BWD) {
                const float v = WL[(l * 32 + ch) * WLD + (cl < C ? cl : 0)];
                w[ks] = cl < C ? v : 0.f;
            } else {
                w[ks] = WL[(l * 32 + cl) * WLD + ch];
            }
        }
    };
    // ---------------- phase 1: forward D step.  A wave owns the column tiles wave and wave + 8; the loads of BOTH are issued before the
    //                  first product (one memory round trip per wave instead of two: the columns come from another XCD's writes)
    {
        constexpr int TPW = (NCT + NW - 1) / NW;
        float va[TPW][KC], vb[TPW][KC];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int t = wave + NW * u;
            const int c = 2 * (t < NCT ? t : 0) + cloc;
            const float *src = a.ws + col_base(c, part, j);
#pragma unroll
            for (int ks = 0; ks < KC; ++ks) {
                const int n = 4 * ks + q;
                const bool in = n <= J, mir = n >= 1 && n <= J;
                va[u][ks] = (a.dbg & 8) ? 1.f : src[(size_t)(in ? n : 0) * pstride];
                vb[u][ks] = (a.dbg & 8) ? 1.f : src[(size_t)(mir ? N0 - n : 0) * pstride];
            }
        }
        // (the weights' trip to LDS rides behind the column loads just issued: one round trip, not two)
        __builtin_amdgcn_sched_barrier(0);
        {
            // all of a thread's (<= 6) weight elements are requested before the first one is stored (a load followed by its LDS
            // store is one exposed round trip per element)
            const float res = a.residual ? 1.f : 0.f;
            constexpr int NWE = (4 * 32 * C + NTH - 1) / NTH;
            float wv[NWE];
            const int nwl = a.L * 32 * C;
#pragma unroll
            for (int e = 0; e < NWE; ++e) {
                const int i = tid + NTH * e;
                const int l = i / (32 * C), rem = i - l * (32 * C), row = rem / C, col = rem - row * C;
                const int ls = i < nwl ? l : 0;          // (elements beyond the last layer read W[0][0]: a.W[l >= L] is NULL)
                const float *Wp = ls == 0 ? a.W[0] : (ls == 1 ? a.W[1] : (ls == 2 ? a.W[2] : a.W[3]));
                wv[e] = Wp[i < nwl ? (row < C ? row : 0) * C + col : 0];
            }
#pragma unroll
            for (int e = 0; e < NWE; ++e) {
                const int i = tid + NTH * e;
                const int l = i / (32 * C), rem = i - l * (32 * C), row = rem / C, col = rem - row * C;
                if (i < nwl) WL[(l * 32 + row) * WLD + col] = row < C ? wv[e] + (row == col ? res : 0.f) : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int t = wave + NW * u;
            if (t >= NCT) break;
            const int c = 2 * t + cloc;
            f32x4 P = {0.f, 0.f, 0.f, 0.f}, Q = P;
            if (!(a.dbg & 1)) {
#pragma unroll
                for (int ks = 0; ks < KC; ++ks) {
                    const int n = 4 * ks + q;
                    const float fa = n <= J ? va[u][ks] : 0.f, fb = (n >= 1 && n <= J) ? vb[u][ks] : 0.f;
                    P = mfma16(tcF[ks], fa + fb, P);
                    Q = mfma16(tsF[ks], fa - fb, Q);
                }
            }
            // accumulator row 4 q + r = k0
            float *dst = PQ + (((c * 2 + part) * 2) * K0) * 4 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = 4 * q + r;
                if (k0 < K0) {
                    dst[k0 * 4] = P[r];
                    dst[(K0 + k0) * 4] = Q[r];
                }
            }
        }
    }
    HNO_STAMP(a.stamps, 1);
    __syncthreads();
    HNO_STAMP(a.stamps, 2);
    if (a.dbg & 32) return;
    // ---------------- phase 2 + 3: Hartley values of the tile's modes, then the layer stack (waves 0 .. NT - 1, one 32-mode tile each)
    const int S0 = 2 * M0, S1 = 2 * m1, S2 = 2 * m2;
    const size_t sample = (size_t)C * S0 * S1 * S2, layer = sample * a.B;
    if (wave < NT) {
        const int mi = wave * 32 + cl;                // mode index: column fastest, then o0, then the sign of (k1, k2)
        const int jm = mi & 3, rest = mi >> 2, s12 = rest / S0, o0 = rest - s12 * S0;
        const int k2 = kt2 * 4 + jm;
        const int kk = o0 < M0 ? o0 : o0 - S0;        // signed k0 of this position
        const int kap = s12 ? -kk : kk;               // the frequency whose X gives the value: H[k] = Re X[k] - Im X[k], H[-k] = Re X[k] + Im X[k]
        const int ka = kap < 0 ? -kap : kap;
        // kept positions along k1 / k2 ([low | high] block): + : (k1, k2), - : (-k1, -k2)
        const int kk1 = s12 ? -k1 : k1;
        const int o1 = kk1 >= 0 ? (kk1 < m1 ? kk1 : -1) : (kk1 >= -m1 ? kk1 + S1 : -1);
        const int o2 = s12 ? (k2 >= 1 && k2 <= m2 ? S2 - k2 : -1) : (k2 < m2 ? k2 : -1);
        const bool valid = mi < NMODE && o1 >= 0 && o2 >= 0;
        const size_t zoff = valid ? ((size_t)o0 * S1 + o1) * S2 + o2 : 0;
        float z[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int c = mid_chan(ks, h);
            const float *pq = PQ + ((c * 2) * 2 * K0 + ka) * 4 + jm;
            const float PR = pq[0], QR = pq[K0 * 4], PI = pq[2 * K0 * 4], QI = pq[3 * K0 * 4];
            const float xr = kap >= 0 ? PR + QI : PR - QI, xi = kap >= 0 ? PI - QR : PI + QR;
            const float hv = a.scale * (s12 ? xr + xi : xr - xi);
            z[ks] = valid ? hv : 0.f;
        }
        float *ob = a.zs + (size_t)b * sample;
        const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
        const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
        const bool lin = a.act == HNO_ACT_NONE;
        float w[NK];
        read_w(BWD ? a.L - 1 : 0, w);      // (written before the barrier behind phase 1)
        if (BWD) {
            // ---- backward of the layer stack on this wave's 32 modes; z holds the incoming gradient (the cropped spectrum of g_u).
            // Saved activations come from zs (z_0 .. z_L; loads unconditional at offset 0 for unkept positions and selected away),
            // one layer ahead of their use; the weight gradient of a layer is a 32 x 32 x 32 product of the G / Z tiles the wave
            // stages through its own LDS (channels become MFMA rows), stored once per layer into the wave's fragment area.
            float *G = GZ + wave * TILE, *Z = G + 32 * GLD;
            const int nslab = a.L * C * C;
            float *mine = MINE + wave * nslab;
            const float *ga = G + l15 * GLD + q, *za = Z + l15 * GLD + q;
            auto load_z = [&](const float *base, float (&v)[NK]) {
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) v[ks] = base[(size_t)mid_chan(ks, h) * (S0 * S1 * S2) + zoff];
            };
            float zo[NK], zi[NK];
            load_z(ob + (size_t)a.L * layer, zo);
            load_z(ob + (size_t)(a.L - 1) * layer, zi);
            HNO_STAMP(a.stamps, 3);
#pragma unroll 1
            for (int l = ((a.dbg & 2) ? -1 : a.L - 1); l >= 0; --l) {
                float zn[NK];
                if (l > 0) load_z(ob + (size_t)(l - 1) * layer, zn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    const int ch = mid_chan(ks, h);
                    const float y = valid ? zo[ks] : 0.f;
                    z[ks] *= (y > 0.f || lin) ? ap : y + aq;
                    G[ch * GLD + cl] = z[ks];
                    Z[ch * GLD + cl] = valid ? zi[ks] : 0.f;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                f32x4 dw[2][2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn) dw[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    float av[2], bv[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        av[m] = ga[m * 16 * GLD + kk * 4];
                        bv[m] = za[m * 16 * GLD + kk * 4];
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nn = 0; nn < 2; ++nn) dw[m][nn] = mfma16(av[m], bv[nn], dw[m][nn]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                float *dst = mine + l * C * C;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int o = m * 16 + q * 4 + r, i = nn * 16 + l15;
                            if (o < C && i < C) dst[o * C + i] = dw[m][nn][r];
                        }
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ks], z[ks], acc, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    z[ks] = acc[ks];
                    zo[ks] = zi[ks];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (l > 0) {
                    read_w(l - 1, w);
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) zi[ks] = zn[ks];
                }
            }
        } else {
        if (valid) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) ob[(size_t)mid_chan(ks, h) * (S0 * S1 * S2) + zoff] = z[ks];
        }
        HNO_STAMP(a.stamps, 3);
#pragma unroll 1
        for (int l = 0; l < ((a.dbg & 2) ? 0 : a.L); ++l) {
            // the next layer's weights are requested before this layer's products and finished (residual identity, zero rows) after
            // its activation: the loads are in flight behind the MFMAs instead of in front of them
            HNO_STAMP(a.stamps, 4 + 4 * l);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ks], z[ks], acc, 0, 0, 0);
            if (a.stamps) asm volatile("s_nop 0" ::"v"(acc[0]));
            HNO_STAMP(a.stamps, 5 + 4 * l);
            if (a.dbg & 512) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = z[r % NK] + w[r % NK];
            }
            ob += layer;
            if (lin) {
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) z[ks] = acc[ks];
            } else {
#pragma unroll
                for (int ks = 0; ks < NK; ks += 2) {
                    const f32x2 y = selu_like_pk(f32x2{acc[ks], acc[ks + 1]}, ap, aq);
                    z[ks] = y[0];
                    z[ks + 1] = y[1];
                }
            }
            if (a.stamps) asm volatile("s_nop 0" ::"v"(z[0]), "v"(z[11]));
            HNO_STAMP(a.stamps, 6 + 4 * l);
            if (valid && !(a.dbg & 256)) {
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) ob[(size_t)mid_chan(ks, h) * (S0 * S1 * S2) + zoff] = z[ks];
            }
            HNO_STAMP(a.stamps, 7 + 4 * l);
            __builtin_amdgcn_sched_barrier(0);
            if (l + 1 < a.L) read_w(l + 1, w);
        }
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) ZL[mid_chan(ks, h) * (NT * 32) + mi] = valid ? z[ks] : 0.f;
    }
    HNO_STAMP(a.stamps, 20);
    __syncthreads();
    HNO_STAMP(a.stamps, 21);
    if (BWD) {   // the NT waves' weight-gradient fragments -> this workgroup's slab (fixed order: bit-reproducible)
        const int nslab = a.L * C * C;
        float *slab = a.partials + (size_t)blockIdx.x * nslab;
        for (int i = tid; i < nslab; i += NTH) {
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < NT; ++wv) sum += MINE[wv * nslab + i];
            slab[i] = sum;
        }
    }
    if (a.dbg & 64) return;
    // ---------------- phase 4: pad + inverse D step
    for (int t = wave; t < NCT; t += NW) {
        if (a.dbg & 4) break;
        const int c = 2 * t + cloc;
        // z(+-, kappa) of this channel and column; positions outside the kept block are zero
        const float *zc = ZL + c * (NT * 32) + j;
        auto zz = [&](int s12, int kap) -> float {
            const bool ok = kap >= -M0 && kap < M0;
            const int o0 = kap >= 0 ? kap : kap + S0;
            const float v = zc[(s12 * S0 + (ok ? o0 : 0)) * 4];
            return ok ? v : 0.f;
        };
        // G'(+k0) = (va + vb) + i (vb - va), G'(-k0) = (vc + vd) + i (vd - vc);  gs = G'(+) + G'(-), gd = G'(+) - G'(-)
        float gs[KI], gd[KI];
#pragma unroll
        for (int ks = 0; ks < KI; ++ks) {
            const int k = 4 * ks + q;
            const bool in = k <= M0;
            const float va = zz(0, k), vb = zz(1, -k), vc = k ? zz(0, -k) : 0.f, vd = k ? zz(1, k) : 0.f;
            const float pr = va + vb, pi = vb - va, mr = vc + vd, mi_ = vd - vc;
            // re part: E = U_re - V_im -> gs = Re sums, gd = Im differences;  im part: E = U_im + V_re
            gs[ks] = in ? (part ? pi + mi_ : pr + mr) : 0.f;
            gd[ks] = in ? (part ? pr - mr : pi - mi_) : 0.f;
        }
        float *dst = a.ws + col_base(c, part, j);
        {   // n = 0: cos = 1, sin = 0 -> the plain sum of the cosine coefficients over k0 = over (ks, q)
            float u0 = gs[0];
#pragma unroll
            for (int ks = 1; ks < KI; ++ks) u0 += gs[ks];
            u0 += __shfl_xor(u0, 16);
            u0 += __shfl_xor(u0, 32);
            if (q == 0) dst[0] = u0;
        }
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            f32x4 U = {0.f, 0.f, 0.f, 0.f}, V = U;
#pragma unroll
            for (int ks = 0; ks < KI; ++ks) {
                U = mfma16(tcI[mt][ks], gs[ks], U);
                V = mfma16(tsI[mt][ks], gd[ks], V);
            }
            // accumulator row 4 q + r <-> n = 1 + 16 mt + 4 q + r;  re part: E[n] = U - V, E[N0 - n] = U + V;  im part: the opposite signs
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 1 + 16 * mt + 4 * q + r;
                const float en = part ? U[r] + V[r] : U[r] - V[r], em = part ? U[r] - V[r] : U[r] + V[r];
                if (n <= J) {      // (the last tile of an N0 with (N0 - 1) / 2 not a multiple of 16 is partly empty)
                    dst[(size_t)n * pstride] = en;
                    if ((N0 & 1) || 2 * n != N0) dst[(size_t)(N0 - n) * pstride] = em;      // (even N0: plane N0 / 2 is its own mirror)
                }
            }
        }
    }
    HNO_STAMP(a.stamps, 22);
}

// ------------------------------------------------------------------------------------------------------------------------
// The same chain for the FOURIER block (FNOSeg: nets/fourier_operator.py:117-223 inside nets/architectures.py:511-608): axis D of the
// rfft + mode selection -> complex channel mix (ONE real 2C x 2C product on [re | im] channels, hno_cmix_compose) -> zero pad + axis D
// of the inverse; and its backward (the transposed D steps around  g <- W2^T g,  dW2 = g s0^T).  Differences to the Hartley kernel:
// the spectrum is the HALF spectrum (k2 in [0, m2)), stored as real data (B, 2C, 2 m0, 2 m1, m2); a (k1, k2) column yields the 2 m0
// complex modes (+-k0, k1, k2) only (no mirrored family); no activation, no residual; the c2r weights (1, 2, 2, ...) over k2 sit on
// the forward D step of the backward and on the inverse D step of the forward.
struct MidFArgs {
    float *ws;                 // as MidArgs.ws
    const float *W2;           // (2C, 2C) composed real form of the complex weights
    float *s0;                 // (B, 2C, 2 m0, 2 m1, m2): forward: written (the cropped spectrum); backward: read
    const float *tw;
    float *partials;           // backward: one (2C x 2C) slab per workgroup
    int B, m1, m2;
    float scale;               // of the forward D step
    int w_fwd, w_inv;          // c2r weights (2 for k2 > 0) on the forward / inverse D step
    int dbg;
    int zl;                    // layout of ws, as MidArgs.zl
};

template <int N0, int M0, bool BWD, bool ZLAY>
__global__ __launch_bounds__(512, 1) void spec_mid_fourier_kernel(MidFArgs a) {
    constexpr int C = 24, C2 = 48, NK = 24, J = N0 / 2, K0 = M0 + 1;
    constexpr int KC = (J + 1 + 3) / 4, KI = (K0 + 3) / 4, NMT = (J + 15) / 16;
    constexpr int S0 = 2 * M0;
    constexpr int NMODE = S0 * 4;                      // (o0, column)
    constexpr int NT = (NMODE + 31) / 32;
    constexpr int NCT = C / 2;
    constexpr int NPQ = C * 2 * 2 * K0 * 4, NZL = C2 * NT * 32;
    constexpr int GLD = 34, TILE = 2 * C2 * GLD;
    static_assert(K0 <= 16 && 2 * M0 <= N0, "one k0 tile");      // (an even N0: the plane N0 / 2 is its own mirror -- see mid_twiddles)
    extern __shared__ float lds[];
    float *PQ = lds, *ZL = lds + NPQ, *GZ = ZL + NZL, *MINE = GZ + NT * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int m1 = a.m1, m2 = a.m2, K1S = 2 * m1 + 1, CP = K1S * 16;
    const int pairs = (K1S + 1) / 2, G8 = (a.B * pairs + 7) & ~7;
    const int w8 = blockIdx.x / G8, gidx = blockIdx.x - w8 * G8;
    const int b = gidx / pairs, k1s = 2 * (gidx - b * pairs) + (w8 >> 2), kt2 = w8 & 3;
    if (b >= a.B || k1s >= K1S) {
        if (BWD)
            for (int i = tid; i < C2 * C2; i += 512) a.partials[(size_t)blockIdx.x * (C2 * C2) + i] = 0.f;
        return;
    }
    const int k1 = k1s - m1;
    const size_t pstride = ZLAY ? (size_t)8 : (size_t)2 * CP;      // (layouts of ws: see spec_mid_kernel)
    auto col_base = [&](int c, int part_, int j_) -> size_t {
        if constexpr (ZLAY) return ((size_t)(k1s * 4 + kt2) * ((size_t)a.B * C * N0) + (size_t)(b * C + c) * N0) * 8 + part_ * 4 + j_;
        return ((size_t)(b * C + c) * N0) * ((size_t)2 * CP) + (size_t)part_ * CP + k1s * 16 + kt2 * 4 + j_;
    };
    const int j = l15 & 3, part = (l15 >> 2) & 1, cloc = l15 >> 3;
    float tcF[KC], tsF[KC], tcI[NMT][KI], tsI[NMT][KI];
    {
        const float *tb = a.tw + lane;
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
            tcF[ks] = tb[ks * 64];
            tsF[ks] = tb[(KC + ks) * 64];
        }
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KI; ++ks) {
                tcI[mt][ks] = tb[(2 * KC + (mt * 2 + 0) * KI + ks) * 64];
                tsI[mt][ks] = tb[(2 * KC + (mt * 2 + 1) * KI + ks) * 64];
            }
    }
    // ---- weights of the channel product as A operands of the 32x32x2 MFMA, two 32-row tiles (rows 48 .. 63 are zero):
    //      forward: lane (row cl, half h), tile ot: W2[32 ot + cl][chan(ks, h)];  backward: the transpose, W2[chan(ks, h)][32 ot + cl].
    //      Rows beyond 2C load row 0 and are selected away (no load under a lane condition, lesson 38).
    const int h = lane >> 5, cl = lane & 31;
    float wA[2][NK];
    if (wave < NT) {
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            const int row = ot * 32 + cl;
            const bool ok = row < C2;
            const int rs = ok ? row : 0;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int ch = mid_chan(ks, h);
                const float v = BWD ? a.W2[ch * C2 + rs] : a.W2[rs * C2 + ch];
                wA[ot][ks] = ok ? v : 0.f;
            }
        }
    }
    // ---- phase 1: forward D step (identical to the Hartley kernel: the plane transform's output does not depend on the layout)
    {
        constexpr int TPW = (NCT + 7) / 8;
        float va[TPW][KC], vb[TPW][KC];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int t = wave + 8 * u;
            const int c = 2 * (t < NCT ? t : 0) + cloc;
            const float *src = a.ws + col_base(c, part, j);
#pragma unroll
            for (int ks = 0; ks < KC; ++ks) {
                const int n = 4 * ks + q;
                const bool in = n <= J, mir = n >= 1 && n <= J;
                va[u][ks] = src[(size_t)(in ? n : 0) * pstride];
                vb[u][ks] = src[(size_t)(mir ? N0 - n : 0) * pstride];
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int t = wave + 8 * u;
            if (t >= NCT) break;
            const int c = 2 * t + cloc;
            f32x4 P = {0.f, 0.f, 0.f, 0.f}, Q = P;
#pragma unroll
            for (int ks = 0; ks < KC; ++ks) {
                const int n = 4 * ks + q;
                const float fa = n <= J ? va[u][ks] : 0.f, fb = (n >= 1 && n <= J) ? vb[u][ks] : 0.f;
                P = mfma16(tcF[ks], fa + fb, P);
                Q = mfma16(tsF[ks], fa - fb, Q);
            }
            float *dst = PQ + (((c * 2 + part) * 2) * K0) * 4 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = 4 * q + r;
                if (k0 < K0) {
                    dst[k0 * 4] = P[r];
                    dst[(K0 + k0) * 4] = Q[r];
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2 + 3: complex spectrum of the tile's modes as 2C real channel values per mode, then the channel product
    const int S1 = 2 * m1;
    const size_t chs = (size_t)S0 * S1 * m2;          // floats per real channel of the spectrum
    if (wave < NT) {
        const int mi = wave * 32 + cl;                // mode index: column fastest, then o0
        const int jm = mi & 3, o0 = mi >> 2;
        const int k2 = kt2 * 4 + jm;
        const int kk = o0 < M0 ? o0 : o0 - S0;        // signed k0 of this position
        const int ka = kk < 0 ? -kk : kk;
        const int o1 = k1 < m1 ? (k1 >= 0 ? k1 : k1 + S1) : -1;      // kept position of k1 in [-m1, m1)
        const bool valid = mi < NMODE && k2 < m2 && o1 >= 0;
        const size_t zoff = valid ? ((size_t)o0 * S1 + o1) * m2 + k2 : 0;
        const float wf = a.scale * ((a.w_fwd && k2 > 0) ? 2.f : 1.f);
        float z[NK];
#pragma unroll
        for (int ks = 0; ks < NK / 2; ++ks) {
            const int c = mid_chan(ks, h);
            const float *pq = PQ + ((c * 2) * 2 * K0 + (valid ? ka : 0)) * 4 + jm;
            const float PR = pq[0], QR = pq[K0 * 4], PI = pq[2 * K0 * 4], QI = pq[3 * K0 * 4];
            // X(+k0) = (PR + QI) + i (PI - QR),  X(-k0) = (PR - QI) + i (PI + QR)
            const float xr = kk >= 0 ? PR + QI : PR - QI, xi = kk >= 0 ? PI - QR : PI + QR;
            z[ks] = valid ? wf * xr : 0.f;            // real channels chan(ks, h)      (ks < 12: the re block)
            z[ks + NK / 2] = valid ? wf * xi : 0.f;   //               chan(ks + 12, h) = C + chan(ks, h): the im block
        }
        float *sb = a.s0 + (size_t)b * C2 * chs;
        f32x16 acc[2];
        if (!BWD) {
            if (valid) {
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) sb[(size_t)mid_chan(ks, h) * chs + zoff] = z[ks];
            }
        } else {
            // weight gradient of the product: dW2[o][i] = sum over modes of g[o] s0[i]; channels become MFMA rows through the wave's LDS
            float zi[NK];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) zi[ks] = sb[(size_t)mid_chan(ks, h) * chs + zoff];
            float *G = GZ + wave * TILE, *Z = G + C2 * GLD;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int ch = mid_chan(ks, h);
                G[ch * GLD + cl] = z[ks];
                Z[ch * GLD + cl] = valid ? zi[ks] : 0.f;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float *ga = G + l15 * GLD + q, *za = Z + l15 * GLD + q;
            f32x4 dw[3][3];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int nn = 0; nn < 3; ++nn) dw[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int kq = 0; kq < 8; ++kq) {
                float av[3], bv[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    av[m] = ga[m * 16 * GLD + kq * 4];
                    bv[m] = za[m * 16 * GLD + kq * 4];
                }
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int nn = 0; nn < 3; ++nn) dw[m][nn] = mfma16(av[m], bv[nn], dw[m][nn]);
            }
            float *mine = MINE + wave * (C2 * C2);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int nn = 0; nn < 3; ++nn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[(m * 16 + q * 4 + r) * C2 + nn * 16 + l15] = dw[m][nn][r];
        }
        // the channel product (forward: s1 = W2 s0; backward: g0 = W2^T g1)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ot][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[ot][ks], z[ks], acc[ot], 0, 0, 0);
        }
        // accumulator register r of tile ot, lane (mode cl, half h): output channel 32 ot + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (ot == 0 || r < 8) ZL[row * (NT * 32) + mi] = valid ? acc[ot][r] : 0.f;
            }
    }
    __syncthreads();
    if (BWD) {
        float *slab = a.partials + (size_t)blockIdx.x * (C2 * C2);
        for (int i = tid; i < C2 * C2; i += 512) {
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < NT; ++wv) sum += MINE[wv * (C2 * C2) + i];
            slab[i] = sum;
        }
    }
    // ---- phase 4: zero pad + inverse D step of column (k1s, 4 kt2 + j), channel c, re / im part
    for (int t = wave; t < NCT; t += 8) {
        const int c = 2 * t + cloc;
        const int k2 = kt2 * 4 + j;
        const float wi = (a.w_inv && k2 > 0) ? 2.f : 1.f;
        auto zz = [&](int isim, int kap) -> float {
            const bool ok = kap >= -M0 && kap < M0;
            const int o0 = kap >= 0 ? kap : kap + S0;
            const float v = ZL[(isim * C + c) * (NT * 32) + (ok ? o0 : 0) * 4 + j];
            return ok ? wi * v : 0.f;
        };
        float gs[KI], gd[KI];
#pragma unroll
        for (int ks = 0; ks < KI; ++ks) {
            const int k = 4 * ks + q;
            const bool in = k <= M0;
            // G'(+k0) = pr + i pi, G'(-k0) = mr + i mi  (positions outside the kept block are zero; ZL holds zeros for unkept columns)
            const float pr = zz(0, k), pi = zz(1, k), mr = k ? zz(0, -k) : 0.f, mi_ = k ? zz(1, -k) : 0.f;
            gs[ks] = in ? (part ? pi + mi_ : pr + mr) : 0.f;
            gd[ks] = in ? (part ? pr - mr : pi - mi_) : 0.f;
        }
        float *dst = a.ws + col_base(c, part, j);
        {
            float u0 = gs[0];
#pragma unroll
            for (int ks = 1; ks < KI; ++ks) u0 += gs[ks];
            u0 += __shfl_xor(u0, 16);
            u0 += __shfl_xor(u0, 32);
            if (q == 0) dst[0] = u0;
        }
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            f32x4 U = {0.f, 0.f, 0.f, 0.f}, V = U;
#pragma unroll
            for (int ks = 0; ks < KI; ++ks) {
                U = mfma16(tcI[mt][ks], gs[ks], U);
                V = mfma16(tsI[mt][ks], gd[ks], V);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 1 + 16 * mt + 4 * q + r;
                const float en = part ? U[r] + V[r] : U[r] - V[r], em = part ? U[r] - V[r] : U[r] + V[r];
                if (n <= J) {      // (the last tile of an N0 with (N0 - 1) / 2 not a multiple of 16 is partly empty)
                    dst[(size_t)n * pstride] = en;
                    if ((N0 & 1) || 2 * n != N0) dst[(size_t)(N0 - n) * pstride] = em;      // (even N0: plane N0 / 2 is its own mirror)
                }
            }
        }
    }
}

// ---- twiddle tables, cached per (device, N0, M0)
static std::map<std::tuple<int, int, int>, float *> g_mid_tw;
static std::mutex g_mid_mutex;

static int mid_twiddles(int N0, int M0, const float **out) {
    int dev = 0;
    HNO_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mid_mutex);
    auto key = std::make_tuple(dev, N0, M0);
    auto it = g_mid_tw.find(key);
    if (it != g_mid_tw.end()) {
        *out = it->second;
        return HNO_OK;
    }
    // A operands of v_mfma_f32_16x16x4_f32 in lane order (lane = 16 q + i holds A[row i][k = q] of its k-step):
    //   forward D step, k-step ks: row i = k0, k = n = 4 ks + q:  cos | sin (2 pi k0 n / N0), zero for k0 > M0 or n > J
    //   inverse D step, tile mt, k-step ks: row i <-> n = 1 + 16 mt + i, k = k0 = 4 ks + q, zero for n > J or k0 > M0
    const int J = N0 / 2, K0 = M0 + 1, KC = (J + 1 + 3) / 4, KI = (K0 + 3) / 4, NMT = (J + 15) / 16;
    std::vector<float> h((size_t)(2 * KC + NMT * 2 * KI) * 64, 0.f);
    const double th = 2.0 * M_PI / N0;
    for (int ks = 0; ks < KC; ++ks)
        for (int ln = 0; ln < 64; ++ln) {
            const int k0 = ln & 15, n = 4 * ks + (ln >> 4);
            if (k0 > M0 || n > J) continue;
            const double ang = th * (double)((long long)k0 * n % N0);
            // (the forward D step folds v[n] + v[N0 - n]: the plane N0 / 2 of an even N0 is its own mirror and would count twice)
            h[(size_t)ks * 64 + ln] = (float)(cos(ang) * (2 * n == N0 ? 0.5 : 1.0));
            h[(size_t)(KC + ks) * 64 + ln] = (float)sin(ang);
        }
    for (int mt = 0; mt < NMT; ++mt)
        for (int ks = 0; ks < KI; ++ks)
            for (int ln = 0; ln < 64; ++ln) {
                const int n = 1 + 16 * mt + (ln & 15), k0 = 4 * ks + (ln >> 4);
                if (k0 > M0 || n > J) continue;
                const double ang = th * (double)((long long)k0 * n % N0);
                h[(size_t)(2 * KC + (mt * 2 + 0) * KI + ks) * 64 + ln] = (float)cos(ang);
                h[(size_t)(2 * KC + (mt * 2 + 1) * KI + ks) * 64 + ln] = (float)sin(ang);
            }
    float *d = nullptr;
    // allocated once per size, outside graph capture (first call = warm-up)
    HNO_CHECK_HIP(hipMalloc((void **)&d, h.size() * sizeof(float)));
    HNO_CHECK_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    g_mid_tw[key] = d;
    *out = d;
    return HNO_OK;
}

// dynamic LDS of spec_mid_kernel (floats): PQ + ZL (+ backward: NT G / Z tiles and NT x L x C x C weight-gradient fragments)
static size_t mid_lds_floats(int M0, int L, bool bwd) {
    const int C = 24, K0 = M0 + 1, NT = (2 * 2 * M0 * 4 + 31) / 32;
    size_t n = (size_t)4 * 32 * 25 + (size_t)C * 2 * 2 * K0 * 4 + (size_t)C * NT * 32;      // layer weights + PQ + ZL
    if (bwd) n += (size_t)NT * 2 * 32 * 34 + (size_t)NT * L * C * C;
    return n;
}

}  // namespace hno

using namespace hno;

// (78: the depth of a 155 x 240 x 240 volume in the reference's (z, y, x) array order, experiments/utils.py:270 -- an even plane count)
// plane counts N0 the fused middle kernels are instantiated for: the working grids of 64^3 ... 256^3 inputs (N0 = size / 2 + 1)
#define HNO_MID_N0_LIST(X) X(33) X(41) X(49) X(57) X(65) X(73) X(78) X(81) X(89) X(97) X(105) X(113) X(121) X(129)
static bool mid_n0_built(int N0) {
#define X(n) if (N0 == n) return true;
    HNO_MID_N0_LIST(X)
#undef X
    return false;
}

// workgroups of the fused middle kernels: 8 x roundup8(B x ceil((2 m1 + 1) / 2)) (spec_mid_kernel's id mapping)
static int mid_workgroups(int B, int m1) {
    const int pairs = (2 * m1 + 1 + 1) / 2, G8 = (B * pairs + 7) & ~7;
    return 8 * G8;
}

// the dynamic-LDS attribute of the backward instantiations is a per-DEVICE property of the loaded code object
static bool mid_attr_needed(int family) {
    static std::mutex m;
    static std::map<std::pair<int, int>, bool> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    std::lock_guard<std::mutex> lock(m);
    bool &d = done[std::make_pair(dev, family)];
    const bool need = !d;
    d = true;
    return need;
}

// slab workspaces of the backward kernels: one slab of weight-gradient partial sums per workgroup (grows with the batch)
extern "C" size_t hno_spec_mid_bwd_workspace_bytes(int B, int C, int m1, int L) {
    return sizeof(float) * (size_t)mid_workgroups(B, m1) * L * C * C;
}
extern "C" size_t hno_spec_mid_fourier_bwd_workspace_bytes(int B, int C, int m1) {
    return sizeof(float) * (size_t)mid_workgroups(B, m1) * 4 * C * C;
}

// 1 if the fused kernels exist for this configuration (Hartley layout, 24 channels, N0 in HNO_MID_N0_LIST with m0 = 10, one k tile per
// plane axis); the caller falls back to hno_dht3_crop / hno_specmix_layers_* / hno_pad_idht3 otherwise.
extern "C" int hno_spec_mid_supported(int C, int N0, int m0, int m1, int m2, int L) {
    return C == 24 && mid_n0_built(N0) && m0 == 10 && 2 * m0 <= N0 && m1 >= 1 && m1 <= 15 && m2 >= 1 && m2 <= 15 && L >= 1 && L <= 4;
}

// workspace: the forward plane transform of x (hno_dht3_planes) on entry, the operand of the inverse plane transform
// (hno_idht3_planes) on return.  zs: (L + 1, B, C, 2 m0, 2 m1, 2 m2), z0 first.
extern "C" int hno_spec_mid_fwd(void *workspace, const float *const *W_layers, float *zs, int B, int C, int N0, int m0, int m1, int m2,
                                int L, int residual, int act, float scale, void *stream) {
    HNO_REQUIRE(workspace && W_layers && zs && B > 0, "hno_spec_mid_fwd: bad argument");
    if (!hno_spec_mid_supported(C, N0, m0, m1, m2, L)) return fail(HNO_ELIMIT, "hno_spec_mid_fwd: unsupported configuration");
    MidArgs a = {};
    a.ws = (float *)workspace;
    for (int l = 0; l < L; ++l) {
        HNO_REQUIRE(W_layers[l], "hno_spec_mid_fwd: W_layers[%d] is NULL", l);
        a.W[l] = W_layers[l];
    }
    a.zs = zs;
    a.B = B;
    a.L = L;
    a.residual = residual;
    a.act = act;
    a.m1 = m1;
    a.m2 = m2;
    a.scale = scale;
    a.dbg = debug_flags();
    a.zl = mid_zlayout() ? 1 : 0;
    a.stamps = (a.dbg & 1024) ? debug_stamp_buffer() : nullptr;
    int rc = mid_twiddles(N0, m0, &a.tw);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int pairs = (2 * m1 + 1 + 1) / 2, G8 = (B * pairs + 7) & ~7;
    const dim3 grid(8 * G8);
    const size_t lds = sizeof(float) * mid_lds_floats(m0, 0, false);
    {
        ProfScope _ps(KID_SPEC_MID_FWD, s, 4.0 * B * C * (2.0 * N0 * 2 * (2 * m1 + 1) * 16 + (L + 1) * 8.0 * m0 * m1 * m2));
#define X(n) if (N0 == n) { if (a.zl) hipLaunchKernelGGL((spec_mid_kernel<n, 10, false, true>), grid, dim3(64 * MID_NW), lds, s, a); else hipLaunchKernelGGL((spec_mid_kernel<n, 10, false, false>), grid, dim3(64 * MID_NW), lds, s, a); }
        HNO_MID_N0_LIST(X)
#undef X
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// Backward of the same chain.  workspace: the forward plane transform of the block-output gradient (hno_dht3_planes of g_u) on entry,
// the operand of the inverse plane transform that yields the block-input gradient (hno_idht3_planes) on return.  zs: what the
// forward wrote (z_0 .. z_L).  dW: (L, C, C) weight gradients (written, or recorded for the batched end-of-backward reduction when
// bit 8 of `residual` is set: hno_set_defer_reduce); slab_workspace: slab_bytes >= hno_spec_mid_bwd_workspace_bytes(B, C, m1, L).
extern "C" int hno_spec_mid_bwd(void *workspace, const float *const *W_layers, const float *zs, float *dW, void *slab_workspace,
                                size_t slab_bytes, int B, int C, int N0, int m0, int m1, int m2, int L, int residual, int act, float scale, void *stream) {
    HNO_REQUIRE(workspace && W_layers && zs && dW && slab_workspace && B > 0, "hno_spec_mid_bwd: bad argument");
    if (!hno_spec_mid_supported(C, N0, m0, m1, m2, L)) return fail(HNO_ELIMIT, "hno_spec_mid_bwd: unsupported configuration");
    MidArgs a = {};
    a.ws = (float *)workspace;
    for (int l = 0; l < L; ++l) {
        HNO_REQUIRE(W_layers[l], "hno_spec_mid_bwd: W_layers[%d] is NULL", l);
        a.W[l] = W_layers[l];
    }
    a.zs = const_cast<float *>(zs);
    a.partials = (float *)slab_workspace;
    a.B = B;
    a.L = L;
    a.residual = residual & 0xff;
    a.act = act;
    a.m1 = m1;
    a.m2 = m2;
    a.scale = scale;
    a.dbg = debug_flags();
    a.zl = mid_zlayout() ? 1 : 0;
    a.stamps = (a.dbg & 1024) ? debug_stamp_buffer() : nullptr;
    int rc = mid_twiddles(N0, m0, &a.tw);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nwg = mid_workgroups(B, m1), n = L * C * C;
    if (slab_bytes < hno_spec_mid_bwd_workspace_bytes(B, C, m1, L))
        return fail(HNO_EINVAL, "hno_spec_mid_bwd: slab workspace of %zu bytes, hno_spec_mid_bwd_workspace_bytes() = %zu", slab_bytes,
                    hno_spec_mid_bwd_workspace_bytes(B, C, m1, L));
    const size_t lds = sizeof(float) * mid_lds_floats(m0, L, true);
    if (mid_attr_needed(0)) {   // once per device
#define X(n) HNO_CHECK_HIP(hipFuncSetAttribute((const void *)spec_mid_kernel<n, 10, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); HNO_CHECK_HIP(hipFuncSetAttribute((const void *)spec_mid_kernel<n, 10, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HNO_MID_N0_LIST(X)
#undef X
    }
    {
        ProfScope _ps(KID_SPEC_MID_BWD, s, 4.0 * B * C * (2.0 * N0 * 2 * (2 * m1 + 1) * 16 + (L + 1) * 8.0 * m0 * m1 * m2));
#define X(n) if (N0 == n) { if (a.zl) hipLaunchKernelGGL((spec_mid_kernel<n, 10, true, true>), dim3(nwg), dim3(64 * MID_NW), lds, s, a); else hipLaunchKernelGGL((spec_mid_kernel<n, 10, true, false>), dim3(nwg), dim3(64 * MID_NW), lds, s, a); }
        HNO_MID_N0_LIST(X)
#undef X
    }
    HNO_CHECK_LAUNCH();
    // bit 8 of residual: record the slab reduction for hno_flush_reduces (per-call form of hno_set_defer_reduce)
    const int prev = hno_set_defer_reduce(0);
    struct Restore { int v; ~Restore() { hno_set_defer_reduce(v); } } restore{prev};
    hno_set_defer_reduce(((residual >> 8) & 1) ? 1 : prev);
    return reduce_partials_launch(a.partials, nwg, n, dW, n, nullptr, s);
}

// ---- Fourier block (spec_mid_fourier_kernel) ---------------------------------------------------------------------------------------
extern "C" int hno_spec_mid_fourier_supported(int C, int N0, int m0, int m1, int m2) {
    return C == 24 && mid_n0_built(N0) && m0 == 10 && 2 * m0 <= N0 && m1 >= 1 && m1 <= 15 && m2 >= 1 && m2 <= 15;
}

static int mid_fourier_launch(bool bwd, void *workspace, const float *W2, float *s0, float *dW2, void *slab_workspace, size_t slab_bytes,
                              int B, int C, int N0,
                              int m0, int m1, int m2, float scale, int w_fwd, int w_inv, void *stream) {
    if (!hno_spec_mid_fourier_supported(C, N0, m0, m1, m2)) return fail(HNO_ELIMIT, "hno_spec_mid_fourier: unsupported configuration");
    MidFArgs a = {};
    a.ws = (float *)workspace;
    a.W2 = W2;
    a.s0 = s0;
    a.partials = (float *)slab_workspace;
    a.B = B;
    a.m1 = m1;
    a.m2 = m2;
    a.scale = scale;
    const bool defer = ((w_fwd >> 8) & 1) != 0;
    a.w_fwd = w_fwd & 0xff;
    a.w_inv = w_inv;
    a.dbg = debug_flags();
    a.zl = mid_zlayout() ? 1 : 0;
    int rc = mid_twiddles(N0, m0, &a.tw);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nwg = mid_workgroups(B, m1), n = 4 * C * C;
    const int K0 = m0 + 1, NT = (2 * m0 * 4 + 31) / 32;
    size_t lds = (size_t)C * 2 * 2 * K0 * 4 + (size_t)2 * C * NT * 32;
    if (bwd) lds += (size_t)NT * 2 * 2 * C * 34 + (size_t)NT * n;
    lds *= sizeof(float);
    if (bwd && slab_bytes < hno_spec_mid_fourier_bwd_workspace_bytes(B, C, m1))
        return fail(HNO_EINVAL, "hno_spec_mid_fourier_bwd: slab workspace of %zu bytes, hno_spec_mid_fourier_bwd_workspace_bytes() = %zu",
                    slab_bytes, hno_spec_mid_fourier_bwd_workspace_bytes(B, C, m1));
    if (bwd && mid_attr_needed(1)) {   // once per device
#define X(n) HNO_CHECK_HIP(hipFuncSetAttribute((const void *)spec_mid_fourier_kernel<n, 10, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); HNO_CHECK_HIP(hipFuncSetAttribute((const void *)spec_mid_fourier_kernel<n, 10, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HNO_MID_N0_LIST(X)
#undef X
    }
    {
        ProfScope _ps(bwd ? KID_SPEC_MID_BWD : KID_SPEC_MID_FWD, s, 4.0 * B * C * (2.0 * N0 * 2 * (2 * m1 + 1) * 16 + 2 * 8.0 * m0 * m1 * m2));
#define X(n)                                                                                                                \
    if (N0 == n && bwd && a.zl) hipLaunchKernelGGL((spec_mid_fourier_kernel<n, 10, true, true>), dim3(nwg), dim3(512), lds, s, a);          \
    else if (N0 == n && bwd) hipLaunchKernelGGL((spec_mid_fourier_kernel<n, 10, true, false>), dim3(nwg), dim3(512), lds, s, a);          \
    else if (N0 == n && a.zl) hipLaunchKernelGGL((spec_mid_fourier_kernel<n, 10, false, true>), dim3(nwg), dim3(512), lds, s, a);          \
    else if (N0 == n) hipLaunchKernelGGL((spec_mid_fourier_kernel<n, 10, false, false>), dim3(nwg), dim3(512), lds, s, a);
        HNO_MID_N0_LIST(X)
#undef X
    }
    HNO_CHECK_LAUNCH();
    if (!bwd) return HNO_OK;
    // bit 8 of w_fwd: record the slab reduction for hno_flush_reduces (the caller then records the real / imaginary split behind it:
    // hno_cmix_split_grad_ex); without it the caller splits dW2 right away: reduce now
    const int prev = hno_set_defer_reduce(0);
    struct Restore { int v; ~Restore() { hno_set_defer_reduce(v); } } restore{prev};
    hno_set_defer_reduce(defer ? 1 : 0);
    return reduce_partials_launch(a.partials, nwg, n, dW2, n, nullptr, s);
}

// workspace: hno_dht3_planes of x on entry, the operand of hno_idht3_planes on return.  W2: (2C, 2C) from hno_cmix_compose.
// s0 (B, 2C, 2 m0, 2 m1, m2): the cropped half spectrum, scaled by `scale` (and by the c2r weights when w_fwd), is written (it is what
// hno_rfft3_crop returns); the mixed spectrum goes straight into the inverse D step (weights w_inv: 1 = the irfft's 1, 2, 2, ...).
extern "C" int hno_spec_mid_fourier_fwd(void *workspace, const float *W2, float *s0, int B, int C, int N0, int m0, int m1, int m2, float scale,
                                        int w_fwd, int w_inv, void *stream) {
    HNO_REQUIRE(workspace && W2 && s0 && B > 0, "hno_spec_mid_fourier_fwd: bad argument");
    return mid_fourier_launch(false, workspace, W2, s0, nullptr, nullptr, 0, B, C, N0, m0, m1, m2, scale, w_fwd, w_inv, stream);
}

// backward: workspace = hno_dht3_planes of the gradient of the inverse transform's output; s0 as written by the forward;
// dW2 (2C, 2C) <- sum over modes of g s0^T (hno_cmix_split_grad turns it into the gradients of the real / imaginary weights);
// slab_workspace: slab_bytes >= hno_spec_mid_fourier_bwd_workspace_bytes(B, C, m1).  On return the workspace is the operand of hno_idht3_planes.
extern "C" int hno_spec_mid_fourier_bwd(void *workspace, const float *W2, const float *s0, float *dW2, void *slab_workspace,
                                        size_t slab_bytes, int B, int C, int N0, int m0, int m1, int m2, float scale, int w_fwd, int w_inv, void *stream) {
    HNO_REQUIRE(workspace && W2 && s0 && dW2 && slab_workspace && B > 0, "hno_spec_mid_fourier_bwd: bad argument");
    return mid_fourier_launch(true, workspace, W2, const_cast<float *>(s0), dW2, slab_workspace, slab_bytes, B, C, N0, m0, m1, m2, scale, w_fwd, w_inv,
                              stream);
}
