// Fused Hartley attention (reference nets/hartley_mha.py:196-201):
//     att[q][k] = act( alpha * sum_c Q[c][q] K[c][k] )          -- SELU by default, NOT softmax: no running normalisation
//     out[c][q] = sum_k V[c][k] att[q][k]
// for every (batch, head) pair, with Q, K (.., Ck, T) and V, out (.., Cv, T), T tokens (1 960 for the published configuration).
// The T x T matrix (61 MB per sample and block in fp32) is never written: forward and backward recompute its tiles on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains -- the 1e-4 parity of the fp32 path).
//
// One templated kernel serves the forward and the two halves of the backward.  A wave OWNS 32 tokens of one side (they sit on
// its lanes for the whole kernel) and STREAMS the other side in tiles of 32 tokens (they sit on accumulator rows):
//     G[s][o] = sum_c Xs[c][s] Xo[c][o]              "score" tile, rows = streamed tokens, lanes = owned tokens
//     acc[c][o] += sum_s Ys[c][s] * f(G)[s][o]        the tile is the B operand of the next MFMA as it lies in the registers
// (a 32 x 32 f32 accumulator has its column on the lane and its rows in the 16 registers, which is exactly the B-operand
// layout of the next 32x32x2 MFMA when that product sums over the ROW index; the k-slot order (i & 3) + 8 (i >> 2) + 4 h of
// register i only has to be matched by the A operand's column index.)
//   MODE 0  forward        owner = queries:  G = S^T (Xs = K, Xo = Q);                         out += V (x) act(G)
//   MODE 1  dQ             owner = queries:  G = S^T, H = dP^T (Xs = V, Xo = dOut);            dQ  += K (x) dS^T
//   MODE 2  dK, dV         owner = keys:     G = S   (Xs = Q, Xo = K), H = dP (Xs = dOut, Xo = V);
//                                                                          dV += dOut (x) act(G);   dK += Q (x) dS
// with dS = alpha * H * act'(G).  The 4 waves of a workgroup own the SAME 32 tokens and split the streamed range; their
// accumulators are added in a fixed order through LDS at the end (no atomics, no workspace).
#include <stdlib.h>
#include <type_traits>

#include "hno_common.h"

namespace hno {

typedef float f32x16h __attribute__((ext_vector_type(16)));

struct HmArgs {
    const float *q, *k, *v, *dout;   // (BZ, Ck, T), (BZ, Ck, T), (BZ, Cv, T), (BZ, Cv, T)
    float *out0, *out1;              // MODE 0: out (BZ, Cv, T); MODE 1: dQ (BZ, Ck, T); MODE 2: dV (out0), dK (out1)
    int BZ, Ck, Cv, T;
    float alpha;
    int act;
    int keep_parts;                  // out0 / out1 are [nsplit] slices of partial results, left unsummed (the consumer adds them in order)
};

// LDS tile of a streamed tensor: [C][33] floats (row c, 32 tokens + 1 pad: reads down a column are conflict-free)
#define HM_LD 33

// stage X[c][s0 .. s0+31] for c < C (rows beyond C up to CP zero) into the wave-private tile; tokens beyond T are zero
__device__ __forceinline__ void hm_stage(float *tile, const float *X, int C, int CP, int T, int s0, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const bool tok = s0 + r < T;
    for (int c = h; c < CP; c += 2) tile[c * HM_LD + r] = (c < C && tok) ? X[(size_t)c * T + s0 + r] : 0.f;
}

// G[s][o] (+)= sum_c tile[c][s] * frag[c]: A operand lane (row s = r, k = h) = tile[2 t + h][r]; B operand = the owner fragment
template <int NSTEP>
__device__ __forceinline__ void hm_score(f32x16h &G, const float *tile, const float (&frag)[NSTEP], int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) G = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(2 * t + h) * HM_LD + r], frag[t], G, 0, 0, 0);
}

// acc[ct][c][o] += sum_s tile[32 ct + c][s] * P[s][o], P in accumulator layout: k-step i takes streamed token
// (i & 3) + 8 (i >> 2) + 4 h from lane half h -- register i of P is the B operand as it is
template <int CT>
__device__ __forceinline__ void hm_accumulate(f32x16h (&acc)[CT], const float *tile, const f32x16h &P, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int s = (i & 3) + 8 * (i >> 2) + 4 * h;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(32 * ct + r) * HM_LD + s], P[i], acc[ct], 0, 0, 0);
    }
}

// accumulator layout: lane column o = r, register i -> channel row (i & 3) + 8 (i >> 2) + 4 h of tile ct.  Every wave writes
// its partial tile into its own LDS region, then the workgroup adds the four regions in a fixed order and stores rows.
template <int CT>
__device__ __forceinline__ void hm_reduce_store(const f32x16h (&acc)[CT], int C, float *dst, const float *smem, float *mine, int stride, int o0,
                                                int T) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) mine[(32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h) * HM_LD + r] = acc[ct][i];
    __syncthreads();
    for (int e = threadIdx.x; e < C * 32; e += 256) {
        const int c = e >> 5, o = e & 31;
        if (o0 + o < T) {
            const float *p = smem + c * HM_LD + o;
            dst[(size_t)c * T + o0 + o] = (p[0] + p[stride]) + (p[2 * stride] + p[3 * stride]);
        }
    }
}

// CKT / CVT: 32-channel tiles of Ck / Cv (channels padded with zeros)
template <int MODE, int CKT, int CVT>
__global__ __launch_bounds__(256) void hmha_kernel(HmArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bz = blockIdx.y;
    const int o0 = blockIdx.x * 32;                       // owned tokens
    const int T = a.T, Ck = a.Ck, Cv = a.Cv;
    constexpr int CKP = 32 * CKT, CVP = 32 * CVT;
    constexpr int CMAX = CKP > CVP ? CKP : CVP;
    float *tile1 = smem + (size_t)wave * 2 * CMAX * HM_LD;   // wave-private: first streamed tensor
    float *tile2 = tile1 + CMAX * HM_LD;                   // second streamed tensor
    const float *Q = a.q + (size_t)bz * Ck * T, *K = a.k + (size_t)bz * Ck * T;
    const float *V = a.v + (size_t)bz * Cv * T, *dO = MODE ? a.dout + (size_t)bz * Cv * T : nullptr;
    // owner fragments: lane (column o = r, k = h) holds X[2 t + h][o0 + r]
    const bool own_ok = o0 + r < T;
    float f1[CKP / 2];                                     // MODE 0, 1: Q; MODE 2: K
    float f2[MODE ? CVP / 2 : 1];                          // MODE 1: dOut; MODE 2: V
    {
        const float *X1 = MODE == 2 ? K : Q;
#pragma unroll
        for (int t = 0; t < CKP / 2; ++t) {
            const int c = 2 * t + h;
            f1[t] = (c < Ck && own_ok) ? X1[(size_t)c * T + o0 + r] : 0.f;
        }
        if constexpr (MODE != 0) {
            const float *X2 = MODE == 2 ? V : dO;
#pragma unroll
            for (int t = 0; t < CVP / 2; ++t) {
                const int c = 2 * t + h;
                f2[t] = (c < Cv && own_ok) ? X2[(size_t)c * T + o0 + r] : 0.f;
            }
        }
    }
    constexpr int CT1 = MODE == 1 ? CKT : CVT;             // channels of the first accumulator (out / dQ / dV)
    f32x16h acc1[CT1], acc2[MODE == 2 ? CKT : 1];
#pragma unroll
    for (int ct = 0; ct < CT1; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[ct][i] = 0.f;
    if constexpr (MODE == 2)
#pragma unroll
        for (int ct = 0; ct < CKT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[ct][i] = 0.f;

    const int ntile = (T + 31) / 32;
    // streamed tensors: MODE 0 / 1: K (scores) and V;  MODE 2: Q (scores) and dOut.  A wave runs alone on its SIMD, so nothing
    // else hides its global latency: the NEXT tile's rows are fetched into registers while the current tile is multiplied
    // (staging and computing in turn ran at 9 % of the fp32 matrix rate).
    const float *S1 = MODE == 2 ? Q : K, *S2 = MODE == 2 ? dO : V;
    float p1[CKP / 2], p2[CVP / 2];
    auto fetch = [&](int s0) {
        const bool tok = s0 + r < T;
#pragma unroll
        for (int t = 0; t < CKP / 2; ++t) {
            const int c = 2 * t + h;
            p1[t] = (c < Ck && tok) ? S1[(size_t)c * T + s0 + r] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < CVP / 2; ++t) {
            const int c = 2 * t + h;
            p2[t] = (c < Cv && tok) ? S2[(size_t)c * T + s0 + r] : 0.f;
        }
    };
    if (wave < ntile) fetch(wave * 32);
    for (int st = wave; st < ntile; st += 4) {              // the 4 waves split the streamed tiles
#pragma unroll
        for (int t = 0; t < CKP / 2; ++t) tile1[(2 * t + h) * HM_LD + r] = p1[t];
#pragma unroll
        for (int t = 0; t < CVP / 2; ++t) tile2[(2 * t + h) * HM_LD + r] = p2[t];
        if (st + 4 < ntile) fetch((st + 4) * 32);
        f32x16h G;
#pragma unroll
        for (int i = 0; i < 16; ++i) G[i] = 0.f;
        hm_score<CKP / 2>(G, tile1, f1, lane);
        f32x16h P;                                          // act(alpha G)
#pragma unroll
        for (int i = 0; i < 16; ++i) P[i] = act_apply(a.alpha * G[i], a.act);
        if constexpr (MODE == 0) {
            hm_accumulate<CVT>(acc1, tile2, P, lane);      // out += V (x) P
        } else {
            f32x16h H;
#pragma unroll
            for (int i = 0; i < 16; ++i) H[i] = 0.f;
            hm_score<CVP / 2>(H, tile2, f2, lane);          // dP = sum_c (V | dOut)[c][s] (dOut | V)[c][o]
            f32x16h dS;
#pragma unroll
            for (int i = 0; i < 16; ++i) dS[i] = a.alpha * H[i] * act_grad_from_out(P[i], a.act);
            if constexpr (MODE == 1) {
                hm_accumulate<CKT>(acc1, tile1, dS, lane);  // dQ += K (x) dS^T
            } else {
                hm_accumulate<CVT>(acc1, tile2, P, lane);   // dV += dOut (x) P
                hm_accumulate<CKT>(acc2, tile1, dS, lane);  // dK += Q (x) dS
            }
        }
    }
    // ---- the 4 waves' partial sums -> LDS (each wave's own tiles are free now), added in order, stored
    if constexpr (MODE == 0) hm_reduce_store<CVT>(acc1, Cv, a.out0 + (size_t)bz * Cv * T, smem, tile1, 2 * CMAX * HM_LD, o0, T);
    else if constexpr (MODE == 1) hm_reduce_store<CKT>(acc1, Ck, a.out0 + (size_t)bz * Ck * T, smem, tile1, 2 * CMAX * HM_LD, o0, T);
    else {
        hm_reduce_store<CVT>(acc1, Cv, a.out0 + (size_t)bz * Cv * T, smem, tile1, 2 * CMAX * HM_LD, o0, T);
        hm_reduce_store<CKT>(acc2, Ck, a.out1 + (size_t)bz * Ck * T, smem, tile1, 2 * CMAX * HM_LD, o0, T);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the same products with SHARED streamed tiles and two waves per SIMD.
//
// hmha_kernel above gives a workgroup's 4 waves the SAME 32 owned tokens and lets each stream its own quarter of the other side
// through a wave-private LDS tile: every wave fetches and stages whole tiles (192 loads + 192 LDS stores per 96 MFMAs), runs alone on
// its SIMD (101 KB of LDS per workgroup) and nothing overlaps its address arithmetic and activation with its MFMAs: 45 TFLOP/s =
// 29 % of the fp32 matrix rate (DESIGN lesson 32).  Here
//   * the 4 waves of a workgroup own DIFFERENT 32-token tiles (128 tokens) and stream the SAME tiles, so a tile is fetched once per
//     workgroup -- by LDS-DMA (global_load_lds_dword: no registers, no LDS store instructions), each wave a quarter of the rows,
//     into the other half of a double buffer while the current tile is multiplied; one barrier per streamed tile;
//   * the streamed range is split over gridDim.y workgroups (8 for the published 4 x 96 x 1960 shape: 2 048 waves = two per SIMD,
//     25 KB of LDS per buffer): each writes a partial result, a small kernel adds the partials in a fixed order (no atomics);
//   * LDS image of a streamed tensor: row PAIRS [c / 2][c & 1][32 tokens] + 1 pad float (one DMA instruction fills a pair); the
//     score product reads along tokens (conflict-free), the accumulation down channels (2-way: pairs share a bank -- 4 LDS cycles
//     per 64-cycle MFMA).
#define HM2_PITCH 65
#define HM_SPLIT_DEFAULT 7      // (bits: MODE 0 / 1 / 2 in split precision)
#define HM2_PITCH4 36
template <int MODE, int CKT, int CVT, bool X4>
__global__ __launch_bounds__(256, MODE == 2 ? 1 : 2) void hmha2_kernel(HmArgs a, float *part0, float *part1) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bz = blockIdx.z, nsplit = gridDim.y, e = blockIdx.y;
    const int o0 = (blockIdx.x * 4 + wave) * 32;            // this wave's owned tokens
    const int T = a.T, Ck = a.Ck, Cv = a.Cv;
    constexpr int CKP = 32 * CKT, CVP = 32 * CVT, NP1 = CKP / 2, NP2 = CVP / 2, NPAIR = NP1 + NP2;
    constexpr int BUF = X4 ? (CKP + CVP) * HM2_PITCH4 : NPAIR * HM2_PITCH;      // floats per buffer
    constexpr int T2OFF = X4 ? CKP * HM2_PITCH4 : NP1 * HM2_PITCH;              // second tensor's image
    constexpr int TSTEP = X4 ? 2 * HM2_PITCH4 : HM2_PITCH;                      // score operand: + t TSTEP
    constexpr int CTSTEP = X4 ? 32 * HM2_PITCH4 : 16 * HM2_PITCH;               // accumulation operand: + ct CTSTEP + s
    const float *Q = a.q + (size_t)bz * Ck * T, *K = a.k + (size_t)bz * Ck * T;
    const float *V = a.v + (size_t)bz * Cv * T, *dO = MODE ? a.dout + (size_t)bz * Cv * T : nullptr;
    // both buffers to zero once: pairs beyond the channel count are never fetched (their products meet zero fragments or land in
    // accumulator rows that are dropped, but 0 x NaN from a previous kernel's LDS contents must not happen)
    for (int i = threadIdx.x; i < 2 * BUF; i += 256) smem[i] = 0.f;
    const bool active = o0 < T;                              // (wave-uniform; idle waves still stage and keep the barriers)
    const bool own_ok = o0 + r < T;
    float f1[CKP / 2];                                      // MODE 0, 1: Q; MODE 2: K
    float f2[MODE ? CVP / 2 : 1];                           // MODE 1: dOut; MODE 2: V
    {
        const float *X1 = MODE == 2 ? K : Q;
#pragma unroll
        for (int t = 0; t < CKP / 2; ++t) {
            const int c = 2 * t + h;
            f1[t] = (c < Ck && own_ok) ? X1[(size_t)c * T + o0 + r] : 0.f;
        }
        if constexpr (MODE != 0) {
            const float *X2 = MODE == 2 ? V : dO;
#pragma unroll
            for (int t = 0; t < CVP / 2; ++t) {
                const int c = 2 * t + h;
                f2[t] = (c < Cv && own_ok) ? X2[(size_t)c * T + o0 + r] : 0.f;
            }
        }
    }
    constexpr int CT1 = MODE == 1 ? CKT : CVT;
    f32x16h acc1[CT1], acc2[MODE == 2 ? CKT : 1];
#pragma unroll
    for (int ct = 0; ct < CT1; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[ct][i] = 0.f;
    if constexpr (MODE == 2)
#pragma unroll
        for (int ct = 0; ct < CKT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[ct][i] = 0.f;

    const int ntile = (T + 31) / 32;
    // branch-free activation (as in the pointwise kernels): act(x) = (x > 0 or linear) ? ap x : aq expm1(x); SELU, ELU or none
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const int st_lo = (int)((long long)ntile * e / nsplit), st_hi = (int)((long long)ntile * (e + 1) / nsplit);
    const float *S1 = MODE == 2 ? Q : K, *S2 = MODE == 2 ? dO : V;
    const unsigned smem_b = (unsigned)(size_t)smem;
    // request tile st into buffer `buf`.
    //  X4 (T % 4 == 0, every row 16-byte aligned): global_load_lds_dwordx4, ONE instruction = 7 rows x (8 data + 1 pad) 16-byte slots
    //     (lane 63 off) = rows of 36 floats in LDS: 14 instructions per 96-row tensor instead of 48 (a DMA instruction costs its wave
    //     60-180 issue cycles beside MFMAs); reads along tokens are conflict-free, reads down channels 4-way (8 banks) -- 8 LDS cycles
    //     per 64-cycle MFMA.  Tokens beyond T re-read the row's last four (their score rows are masked below).
    //  else: global_load_lds_dword, one instruction = a row PAIR [c / 2][c & 1][32 tokens] + 1 pad float; the odd row of a last half
    //     pair re-reads its even row (dropped / zero-fragment rows).
    const int l9 = lane / 9, q9 = lane - 9 * l9;
    auto request = [&](int st, int buf) {
        if constexpr (X4) {
            const int tok = st * 32 + 4 * (q9 < 8 ? q9 : 7);
            const unsigned tokoff = (unsigned)(tok + 4 <= T ? tok : T - 4) * 4u;
            constexpr int NI1 = (CKP + 6) / 7, NI2 = (CVP + 6) / 7;
            for (int n = wave; n < NI1 + NI2; n += 4) {
                const bool first = n < NI1;                      // wave-uniform
                const int n7 = 7 * (first ? n : n - NI1), c = n7 + l9, C = first ? Ck : Cv;
                const float *base = first ? S1 : S2;
                const unsigned off = (unsigned)(c < C ? c : 0) * (unsigned)T * 4u + tokoff;
                const unsigned dst = __builtin_amdgcn_readfirstlane(smem_b + (unsigned)(buf * BUF + (first ? 0 : T2OFF) + n7 * HM2_PITCH4) * 4u);
                if (lane < 63 && c < C)
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(dst) : "memory");
            }
        } else {
            const int tokc = st * 32 + r < T ? st * 32 + r : T - 1;
            for (int j = wave; j < NPAIR; j += 4) {
                const bool first = j < NP1;
                const int c0 = 2 * (first ? j : j - NP1), C = first ? Ck : Cv;
                if (c0 >= C) continue;                           // wave-uniform
                const float *base = (first ? S1 : S2) + (size_t)c0 * T;
                const unsigned off = (unsigned)(((c0 + 1 < C && h) ? T : 0) + tokc) * 4u;
                dma_row_pair(base, off, __builtin_amdgcn_readfirstlane(smem_b + (unsigned)(buf * BUF + j * HM2_PITCH) * 4u));
            }
        }
    };
    __syncthreads();                                         // the zero fill is done before the first DMA lands
    if (st_lo < st_hi) request(st_lo, 0);
    for (int st = st_lo; st < st_hi; ++st) {
        const int buf = (st - st_lo) & 1;
        dma_wait<0>();                                       // this wave's pieces of tile st have landed ...
        __syncthreads();                                     // ... everybody's have, and everybody is done with tile st - 1
        if (st + 1 < st_hi) request(st + 1, buf ^ 1);
        if (!active) continue;
        const float *t1 = smem + buf * BUF, *t2 = t1 + T2OFF;
        // score operands (+ t TSTEP): channel 2 t + h, token r;  accumulation operands (+ ct CTSTEP + s): channel 32 ct + r, token s
        const int rowo = X4 ? h * HM2_PITCH4 + r : h * 32 + r;
        const int colo = X4 ? r * HM2_PITCH4 : (r >> 1) * HM2_PITCH + (r & 1) * 32;
        const float *row1 = t1 + rowo, *row2 = t2 + rowo, *col1 = t1 + colo, *col2 = t2 + colo;
        // LDS operands are requested a chunk ahead of the MFMAs that use them (the compiler's own order was read, wait, two MFMAs,
        // read, wait, ...: an exposed LDS round trip per pair)
        auto score = [&](f32x16h &Gm, const float *row, const float *frag, auto nn) {
            constexpr int N = decltype(nn)::value, CH = 8, NCH = N / CH;
            static_assert(N % CH == 0, "whole chunks");
            float cur[CH], nxt[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) cur[u] = row[u * TSTEP];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c + 1 < NCH) {
#pragma unroll
                    for (int u = 0; u < CH; ++u) nxt[u] = row[((c + 1) * CH + u) * TSTEP];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; ++u) Gm = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[u], frag[c * CH + u], Gm, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; ++u) cur[u] = nxt[u];
            }
        };
        f32x16h G;
#pragma unroll
        for (int i = 0; i < 16; ++i) G[i] = 0.f;
        score(G, row1, f1, std::integral_constant<int, CKP / 2>{});
        const int sbase = st * 32 + 4 * h;
        f32x16h P;                                           // act(alpha G), rows of tokens beyond T zero
        if (lin) {
#pragma unroll
            for (int i = 0; i < 16; ++i) P[i] = sbase + (i & 3) + 8 * (i >> 2) < T ? a.alpha * G[i] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 y = selu_like_pk(f32x2{a.alpha * G[i], a.alpha * G[i + 1]}, ap, aq);
                P[i] = sbase + (i & 3) + 8 * (i >> 2) < T ? y[0] : 0.f;
                P[i + 1] = sbase + ((i + 1) & 3) + 8 * ((i + 1) >> 2) < T ? y[1] : 0.f;
            }
        }
        auto accumulate = [&](f32x16h *acc, const float *col, const f32x16h &Pm, auto ctn) {
            constexpr int CT = decltype(ctn)::value, IS = 2;         // chunks of IS k-steps (IS x CT operands)
            float cur[IS][CT], nxt[IS][CT];
            auto ld = [&](float (&d)[IS][CT], int i0) {
#pragma unroll
                for (int u = 0; u < IS; ++u) {
                    const int i = i0 + u, sidx = (i & 3) + 8 * (i >> 2) + 4 * h;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) d[u][ct] = col[ct * CTSTEP + sidx];
                }
            };
            ld(cur, 0);
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += IS) {
                if (i0 + IS < 16) ld(nxt, i0 + IS);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < IS; ++u)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[u][ct], Pm[i0 + u], acc[ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < IS; ++u)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) cur[u][ct] = nxt[u][ct];
            }
        };
        if constexpr (MODE == 0) {
            accumulate(acc1, col2, P, std::integral_constant<int, CVT>{});          // out += V (x) P
        } else {
            f32x16h H;
#pragma unroll
            for (int i = 0; i < 16; ++i) H[i] = 0.f;
            score(H, row2, f2, std::integral_constant<int, CVP / 2>{});
            f32x16h dS;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // act'(x) from the output y = act(x): SELU-like: y > 0 ? ap : y + aq (branch-free; masked rows have P = 0, H finite)
                const float gr = (lin || P[i] > 0.f) ? ap : P[i] + aq;
                const float d = a.alpha * H[i] * gr;
                dS[i] = sbase + (i & 3) + 8 * (i >> 2) < T ? d : 0.f;
            }
            if constexpr (MODE == 1) {
                accumulate(acc1, col1, dS, std::integral_constant<int, CKT>{});     // dQ += K (x) dS^T
            } else {
                accumulate(acc1, col2, P, std::integral_constant<int, CVT>{});      // dV += dOut (x) P
                accumulate(acc2, col1, dS, std::integral_constant<int, CKT>{});     // dK += Q (x) dS
            }
        }
    }
    // ---- this wave's partial tiles: register i of tile ct = channel 32 ct + (i & 3) + 8 (i >> 2) + 4 h, lane column = token o0 + r
    if (active && own_ok) {
        const int C1 = MODE == 1 ? Ck : Cv;
        float *d0 = part0 + (((size_t)e * a.BZ + bz) * C1) * T + o0 + r;
#pragma unroll
        for (int ct = 0; ct < CT1; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (c < C1) d0[(size_t)c * T] = acc1[ct][i];
            }
        if constexpr (MODE == 2) {
            float *d1 = part1 + (((size_t)e * a.BZ + bz) * Ck) * T + o0 + r;
#pragma unroll
            for (int ct = 0; ct < CKT; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (c < Ck) d1[(size_t)c * T] = acc2[ct][i];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 6: the same products in SPLIT PRECISION on the bf16 matrix cores.
//
// hmha2_kernel spends 6 144 matrix-pipe cycles per 32 x 32 score tile on v_mfma_f32_32x32x2_f32 (64 cycles for 4 K flops).  An fp32
// value is x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) -- 24 significant bits -- and a product
//     a b = a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0 + O(2^-24 |a b|)
// is six v_mfma_f32_32x32x16_bf16 (32 cycles for 32 K flops each, fp32 accumulation; every bf16 x bf16 product is exact in fp32): 2.7x
// fewer matrix cycles for fp32-level results (a timing probe with these instruction counts took the HartleyMHASeg step from 10.1 to
// 7.9 ms before the cost of the splits: LESSONS 101).  The splits are VALU work, so the STREAMED tiles are split ONCE per workgroup --
// the four waves used to read the same fp32 tile -- by the threads that stage them: global -> registers (a tile ahead of the products) ->
// three bf16 planes in LDS, in the layout each product reads with 16- / 8-byte LDS loads:
//     TK[term][token][channel]   the score product's A operand (8 consecutive channels of a token per lane)
//     CT[term][channel][token]   the accumulation's A operand (2 x 4 consecutive tokens of a channel per lane, in the accumulator's
//                                k-slot order (i & 3) + 8 (i >> 2) + 4 h, so that the score tile P is again the B operand as it lies
//                                in the registers -- split per lane)
// The owner's fragments are split once in the prologue.  T % 4 == 0 (16-byte token quads).
typedef __bf16 hbf8 __attribute__((ext_vector_type(8)));
struct HmT3 { uint4 t[3]; };            // three bf16 terms of eight values

// bf16 terms by integer arithmetic.  The first two terms TRUNCATE (x & 0xffff0000: the term is an fp32 number, so the remainder x - x0
// is exact, and the halves of a pair are packed by ONE v_perm_b32 straight from the unmasked words); the third rounds what is left
// (half up in magnitude): |x - (x0 + x1 + x2)| <= 2^-23 |x|, unbiased.
// ~6.5 instructions per value; the v_cvt / shift / pack sequence of the first version was ~12 and made the kernel VALU-bound.
__device__ __forceinline__ void hm_split_pair(float a, float b, unsigned &p0, unsigned &p1, unsigned &p2) {
    const unsigned M = 0xffff0000u;
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float fa = a - __builtin_bit_cast(float, ua & M), fb = b - __builtin_bit_cast(float, ub & M);
    const unsigned ra = __builtin_bit_cast(unsigned, fa), rb = __builtin_bit_cast(unsigned, fb);
    p1 = __builtin_amdgcn_perm(rb, ra, 0x07060302u);
    const float qa = fa - __builtin_bit_cast(float, ra & M), qb = fb - __builtin_bit_cast(float, rb & M);
    p2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, qb) + 0x8000u, __builtin_bit_cast(unsigned, qa) + 0x8000u, 0x07060302u);
}
__device__ __forceinline__ HmT3 hm_split8(const float (&v)[8]) {
    HmT3 o;
    hm_split_pair(v[0], v[1], o.t[0].x, o.t[1].x, o.t[2].x);
    hm_split_pair(v[2], v[3], o.t[0].y, o.t[1].y, o.t[2].y);
    hm_split_pair(v[4], v[5], o.t[0].z, o.t[1].z, o.t[2].z);
    hm_split_pair(v[6], v[7], o.t[0].w, o.t[1].w, o.t[2].w);
    return o;
}
// D (+)= A B in split precision: two accumulators (the six products alternate) keep two independent chains in the matrix pipe
__device__ __forceinline__ void hm_mfma6(f32x16h &d0, f32x16h &d1, const HmT3 &a, const HmT3 &b) {
#define HM_BF(x) __builtin_bit_cast(hbf8, x)
    d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(HM_BF(a.t[0]), HM_BF(b.t[0]), d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(HM_BF(a.t[0]), HM_BF(b.t[1]), d1, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(HM_BF(a.t[1]), HM_BF(b.t[0]), d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(HM_BF(a.t[0]), HM_BF(b.t[2]), d1, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(HM_BF(a.t[1]), HM_BF(b.t[1]), d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(HM_BF(a.t[2]), HM_BF(b.t[0]), d1, 0, 0, 0);
#undef HM_BF
}

template <int MODE, int CKT, int CVT>
__global__ __launch_bounds__(256, MODE == 0 ? 2 : 1) void hmha3_kernel(HmArgs a, float *part0, float *part1) {
    extern __shared__ unsigned short sm16[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bz = blockIdx.z, nsplit = gridDim.y, e = blockIdx.y;
    const int o0 = (blockIdx.x * 4 + wave) * 32;            // this wave's owned tokens
    const int T = a.T, Ck = a.Ck, Cv = a.Cv;
    constexpr int CKP = 32 * CKT, CVP = 32 * CVT;
    constexpr int NJK = CKP / 16, NJV = CVP / 16;           // 16-channel chunks of a score product
    // LDS planes (bf16 elements).  TK rows: CP + 8 (16-byte row alignment, rows 2 banks apart per 8 channels); CT rows: 32 + 8
    constexpr int PK1 = CKP + 8, PV1 = CVP + 8, P2 = 40;
    // which planes a mode keeps:            S1 (K / K / Q)              S2 (V / V / dO)
    constexpr bool S1_TK = true, S1_CT = MODE != 0, S2_TK = MODE != 0, S2_CT = MODE != 1;
    constexpr int O_S1TK = 0;
    constexpr int O_S1CT = O_S1TK + (S1_TK ? 3 * 32 * PK1 : 0);
    constexpr int O_S2TK = O_S1CT + (S1_CT ? 3 * CKP * P2 : 0);
    constexpr int O_S2CT = O_S2TK + (S2_TK ? 3 * 32 * PV1 : 0);
    const float *Q = a.q + (size_t)bz * Ck * T, *K = a.k + (size_t)bz * Ck * T;
    const float *V = a.v + (size_t)bz * Cv * T, *dO = MODE ? a.dout + (size_t)bz * Cv * T : nullptr;
    const bool active = o0 < T;                              // (wave-uniform; idle waves still stage and keep the barriers)
    const bool own_ok = o0 + r < T;
    // ---- the owner's fragments, split once: chunk j, lane half h: channels 16 j + 8 h .. + 7 of token o0 + r
    HmT3 f1[NJK], f2[MODE ? NJV : 1];
    {
        const float *X1 = MODE == 2 ? K : Q;
#pragma unroll
        for (int j = 0; j < NJK; ++j) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = 16 * j + 8 * h + u;
                v[u] = (c < Ck && own_ok) ? X1[(size_t)c * T + o0 + r] : 0.f;
            }
            f1[j] = hm_split8(v);
        }
        if constexpr (MODE != 0) {
            const float *X2 = MODE == 2 ? V : dO;
#pragma unroll
            for (int j = 0; j < NJV; ++j) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = 16 * j + 8 * h + u;
                    v[u] = (c < Cv && own_ok) ? X2[(size_t)c * T + o0 + r] : 0.f;
                }
                f2[j] = hm_split8(v);
            }
        }
    }
    constexpr int CT1 = MODE == 1 ? CKT : CVT;
    f32x16h acc1[CT1], acc2[MODE == 2 ? CKT : 1];      // (one chain per channel tile: the tiles' chains interleave in the matrix pipe)
#pragma unroll
    for (int ct = 0; ct < CT1; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[ct][i] = 0.f;
    if constexpr (MODE == 2)
#pragma unroll
        for (int ct = 0; ct < CKT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[ct][i] = 0.f;

    const int ntile = (T + 31) / 32;
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const int st_lo = (int)((long long)ntile * e / nsplit), st_hi = (int)((long long)ntile * (e + 1) / nsplit);
    const float *S1 = MODE == 2 ? Q : K, *S2 = MODE == 2 ? dO : V;
    // ---- staging.  TK shape: thread (token = tid & 31, q = tid >> 5) takes 4 CT channels of its token; CT shape: item = (channel,
    // token quad), CT items per thread.  Tokens beyond T re-read valid ones (their score rows are masked below).
    constexpr int NV1 = 4 * CKT, NV2 = 4 * CVT;
    float g1tk[S1_TK ? NV1 : 1], g2tk[S2_TK ? NV2 : 1];
    float4 g1ct[S1_CT ? CKT : 1], g2ct[S2_CT ? CVT : 1];
    const int tkt = threadIdx.x & 31, tkq = threadIdx.x >> 5;
    // (32-bit byte offsets from the tensors' bases: a row's offset u T 4 is wave-uniform, the per-lane part is one add per load; the
    // 64-bit index arithmetic of the first version was ~100 of the ~900 vector instructions per tile)
    const unsigned Tb = (unsigned)T * 4u;
    auto fetch = [&](int st) {
        const unsigned tok = (unsigned)(st * 32 + tkt < T ? st * 32 + tkt : T - 1);
        if constexpr (S1_TK) {
            const unsigned o = (unsigned)(NV1 * tkq) * Tb + tok * 4u;
#pragma unroll
            for (int u = 0; u < NV1; ++u) g1tk[u] = NV1 * tkq + u < Ck ? ld_off(S1, o + (unsigned)u * Tb) : 0.f;
        }
        if constexpr (S2_TK) {
            const unsigned o = (unsigned)(NV2 * tkq) * Tb + tok * 4u;
#pragma unroll
            for (int u = 0; u < NV2; ++u) g2tk[u] = NV2 * tkq + u < Cv ? ld_off(S2, o + (unsigned)u * Tb) : 0.f;
        }
        const int q4 = threadIdx.x & 7;
        const unsigned tk = (unsigned)(st * 32 + 4 * q4 + 4 <= T ? st * 32 + 4 * q4 : T - 4);
        if constexpr (S1_CT) {
#pragma unroll
            for (int n = 0; n < CKT; ++n) {
                const int c = (threadIdx.x >> 3) + 32 * n;
                g1ct[n] = c < Ck ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(S1) + ((unsigned)c * Tb + tk * 4u)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (S2_CT) {
#pragma unroll
            for (int n = 0; n < CVT; ++n) {
                const int c = (threadIdx.x >> 3) + 32 * n;
                g2ct[n] = c < Cv ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(S2) + ((unsigned)c * Tb + tk * 4u)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto put_tk = [&](const float *g, int NV, int base, int pitch) {
        // NV consecutive channels of token tkt -> the three planes, 4-byte pairs
#pragma unroll
        for (int u = 0; u < NV; u += 2) {
            unsigned p0, p1, p2;
            hm_split_pair(g[u], g[u + 1], p0, p1, p2);
            const int off = base + tkt * pitch + NV * tkq + u;
            *reinterpret_cast<unsigned *>(sm16 + off) = p0;
            *reinterpret_cast<unsigned *>(sm16 + off + 32 * pitch) = p1;
            *reinterpret_cast<unsigned *>(sm16 + off + 64 * pitch) = p2;
        }
    };
    auto put_ct = [&](const float4 *g, int CTn, int CP, int base) {
#pragma unroll
        for (int n = 0; n < CTn; ++n) {
            const int item = threadIdx.x + 256 * n, c = item >> 3, q4 = item & 7;
            unsigned a0, a1, a2, b0, b1, b2;
            hm_split_pair(g[n].x, g[n].y, a0, a1, a2);
            hm_split_pair(g[n].z, g[n].w, b0, b1, b2);
            const int off = base + c * P2 + 4 * q4;
            *reinterpret_cast<uint2 *>(sm16 + off) = make_uint2(a0, b0);
            *reinterpret_cast<uint2 *>(sm16 + off + CP * P2) = make_uint2(a1, b1);
            *reinterpret_cast<uint2 *>(sm16 + off + 2 * CP * P2) = make_uint2(a2, b2);
        }
    };
    auto store = [&]() {
        if constexpr (S1_TK) put_tk(g1tk, NV1, O_S1TK, PK1);
        if constexpr (S2_TK) put_tk(g2tk, NV2, O_S2TK, PV1);
        if constexpr (S1_CT) put_ct(g1ct, CKT, CKP, O_S1CT);
        if constexpr (S2_CT) put_ct(g2ct, CVT, CVP, O_S2CT);
    };
    // operand fetches
    auto tk_operand = [&](int base, int pitch, int j) {      // score A operand of chunk j: streamed token r, channels 16 j + 8 h .. + 7
        HmT3 o;
        const int off = base + r * pitch + 16 * j + 8 * h;
#pragma unroll
        for (int t = 0; t < 3; ++t) o.t[t] = *reinterpret_cast<const uint4 *>(sm16 + off + t * 32 * pitch);
        return o;
    };
    auto ct_operand = [&](int base, int CP, int ct, int j) {  // accumulation A operand: channel 32 ct + r, tokens 16 j + 4 h + {0..3, 8..11}
        HmT3 o;
        const int off = base + (32 * ct + r) * P2 + 16 * j + 4 * h;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(sm16 + off + t * CP * P2), hi = *reinterpret_cast<const uint2 *>(sm16 + off + t * CP * P2 + 8);
            o.t[t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        return o;
    };
    auto split_chunk = [&](const f32x16h &Pm, int j) {    // chunk j = registers 8 j .. 8 j + 7 (the accumulator's own k-slot order)
        const float v[8] = {Pm[8 * j], Pm[8 * j + 1], Pm[8 * j + 2], Pm[8 * j + 3], Pm[8 * j + 4], Pm[8 * j + 5], Pm[8 * j + 6], Pm[8 * j + 7]};
        return hm_split8(v);
    };

    if (st_lo < st_hi) {
        fetch(st_lo);
        store();
    }
    __syncthreads();
    for (int st = st_lo; st < st_hi; ++st) {
        if (st + 1 < st_hi) fetch(st + 1);                   // global loads in flight during this tile's products
        if (active) {
            f32x16h G0;
#pragma unroll
            for (int i = 0; i < 16; ++i) G0[i] = 0.f;
#pragma unroll
            for (int j = 0; j < NJK; ++j) {
                const HmT3 av = tk_operand(O_S1TK, PK1, j);
                hm_mfma6(G0, G0, av, f1[j]);
            }
            const int sbase = st * 32 + 4 * h;
            const bool tail = st * 32 + 32 > T;              // (uniform) only the last tile has rows of tokens beyond T: masked to zero
            f32x16h P;                                       // act(alpha G)
            if (lin) {
#pragma unroll
                for (int i = 0; i < 16; ++i) P[i] = a.alpha * G0[i];
            } else {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 y = selu_like_pk(f32x2{a.alpha * G0[i], a.alpha * G0[i + 1]}, ap, aq);
                    P[i] = y[0];
                    P[i + 1] = y[1];
                }
            }
            if (tail) {
#pragma unroll
                for (int i = 0; i < 16; ++i) P[i] = sbase + (i & 3) + 8 * (i >> 2) < T ? P[i] : 0.f;
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const HmT3 Pj = split_chunk(P, j);
#pragma unroll
                    for (int ct = 0; ct < CVT; ++ct) hm_mfma6(acc1[ct], acc1[ct], ct_operand(O_S2CT, CVP, ct, j), Pj);      // out += V (x) P
                }
            } else {
                f32x16h H0;
#pragma unroll
                for (int i = 0; i < 16; ++i) H0[i] = 0.f;
#pragma unroll
                for (int j = 0; j < NJV; ++j) {
                    const HmT3 av = tk_operand(O_S2TK, PV1, j);
                    hm_mfma6(H0, H0, av, f2[j]);
                }
                f32x16h dS;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float gr = (lin || P[i] > 0.f) ? ap : P[i] + aq;
                    dS[i] = a.alpha * H0[i] * gr;
                }
                if (tail) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) dS[i] = sbase + (i & 3) + 8 * (i >> 2) < T ? dS[i] : 0.f;
                }
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const HmT3 Dj = split_chunk(dS, j);
#pragma unroll
                        for (int ct = 0; ct < CKT; ++ct) hm_mfma6(acc1[ct], acc1[ct], ct_operand(O_S1CT, CKP, ct, j), Dj);  // dQ += K (x) dS^T
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const HmT3 Pj = split_chunk(P, j);
#pragma unroll
                        for (int ct = 0; ct < CVT; ++ct) hm_mfma6(acc1[ct], acc1[ct], ct_operand(O_S2CT, CVP, ct, j), Pj);  // dV += dOut (x) P
                        const HmT3 Dj = split_chunk(dS, j);
#pragma unroll
                        for (int ct = 0; ct < CKT; ++ct) hm_mfma6(acc2[ct], acc2[ct], ct_operand(O_S1CT, CKP, ct, j), Dj);  // dK += Q (x) dS
                    }
                }
            }
        }
        __syncthreads();                                     // every wave has read tile st
        if (st + 1 < st_hi) store();
        __syncthreads();                                     // tile st + 1 is in place
    }
    // ---- this wave's partial tiles (hmha2_kernel's layout)
    if (active && own_ok) {
        const int C1 = MODE == 1 ? Ck : Cv;
        float *d0 = part0 + (((size_t)e * a.BZ + bz) * C1) * T + o0 + r;
#pragma unroll
        for (int ct = 0; ct < CT1; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (c < C1) d0[(size_t)c * T] = acc1[ct][i];
            }
        if constexpr (MODE == 2) {
            float *d1 = part1 + (((size_t)e * a.BZ + bz) * Ck) * T + o0 + r;
#pragma unroll
            for (int ct = 0; ct < CKT; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (c < Ck) d1[(size_t)c * T] = acc2[ct][i];
                }
        }
    }
}

template <int MODE, int CKT, int CVT>
static size_t hm3_lds_bytes() {
    constexpr int CKP = 32 * CKT, CVP = 32 * CVT, PK1 = CKP + 8, PV1 = CVP + 8, P2 = 40;
    size_t n = 3 * 32 * PK1;                        // S1 TK
    if (MODE != 0) n += 3 * CKP * P2 + 3 * 32 * PV1; // S1 CT, S2 TK
    if (MODE != 1) n += 3 * CVP * P2;                // S2 CT
    return n * 2;
}

// out[i] = part[0][i] + part[1][i] + ... (fixed order), n floats per partial
__global__ __launch_bounds__(256) void hm_sum_partials_kernel(const float *__restrict__ part, float *__restrict__ out, long long n, int nsplit) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = part[i];
        for (int e = 1; e < nsplit; ++e) s += part[(size_t)e * n + i];
        out[i] = s;
    }
}

template <int MODE, int CKT, int CVT, bool X4>
static int hm2_launch_x(const HmArgs &a, float *workspace, hipStream_t s);

// stream splits so that the launch has `waves` waves (2 048 = two per SIMD) without making a split shorter than two tiles
static int hm2_nsplit(int BZ, int T, int waves = 2048) {
    const int nblk = (T + 127) / 128, ntile = (T + 31) / 32;
    static const int force = getenv("HNO_HM_NSPLIT") ? atoi(getenv("HNO_HM_NSPLIT")) : 0;     // tuning aid
    int n = (waves + nblk * 4 * BZ - 1) / (nblk * 4 * BZ);
    if (n > 8) n = 8;
    if (force > 0 && force <= 16) n = force;
    if (n > ntile / 2) n = ntile / 2;
    return n < 1 ? 1 : n;
}
static int hm_split_modes() {
    static const int m = getenv("HNO_HM_SPLIT") ? atoi(getenv("HNO_HM_SPLIT")) : HM_SPLIT_DEFAULT;
    return m;
}
// the split-precision kernels serve this shape in this mode (hm2_launch_x)
static bool hm3_serves(int mode, int Ck, int Cv, int T) {
    const int kt = (Ck + 31) / 32, vt = (Cv + 31) / 32;
    const bool same = (kt == 1 && vt == 1) || (kt <= 2 && vt <= 2) || (kt == 3 && vt == 3);      // hm_dispatch's instantiations up to 3 tiles
    return same && T % 4 == 0 && T >= 4 && !(debug_flags() & HNO_DBG_HM_PAIR) && ((hm_split_modes() >> mode) & 1);
}
// The BACKWARD split-precision kernels hold one workgroup per CU (362 / 483 registers per lane): 8 splits of the published shape are 512
// workgroups = two rounds, each with its own prologue (the owner's fragments loaded and split) and twice the partial sums for the
// ungrouping kernel to add.  One round (1 024 waves): dQ 89 -> 81 us, dK / dV 114 -> 105, the summing ungrouping 13.2 -> 10.4 (LESSONS
// 107); the forward kernel (two workgroups per CU) keeps its 2 048.  HNO_HM_NSPLIT_BWD=0: as the forward.
static int hm_nsplit_for(int mode, int BZ, int Ck, int Cv, int T) {
    static const bool bwd_one_round = !(getenv("HNO_HM_NSPLIT_BWD") && atoi(getenv("HNO_HM_NSPLIT_BWD")) == 0);
    if (mode != 0 && bwd_one_round && hm3_serves(1, Ck, Cv, T) && hm3_serves(2, Ck, Cv, T)) return hm2_nsplit(BZ, T, 1024);
    return hm2_nsplit(BZ, T);
}

template <int MODE, int CKT, int CVT>
static int hm2_launch(const HmArgs &a, float *workspace, hipStream_t s) {
    if (a.T % 4 == 0 && a.T >= 4 && !(debug_flags() & HNO_DBG_HM_PAIR)) return hm2_launch_x<MODE, CKT, CVT, true>(a, workspace, s);
    return hm2_launch_x<MODE, CKT, CVT, false>(a, workspace, s);
}

template <int MODE, int CKT, int CVT, bool X4>
static int hm2_launch_x(const HmArgs &a, float *workspace, hipStream_t s) {
    constexpr int NPAIR = 16 * (CKT + CVT);
    const size_t lds = (size_t)2 * (X4 ? 32 * (CKT + CVT) * HM2_PITCH4 : NPAIR * HM2_PITCH) * sizeof(float);
    static int attr = -1;
    if (lds > 48 * 1024 && attr != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)hmha2_kernel<MODE, CKT, CVT, X4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = current_device();
    }
    const int nsplit = hm_nsplit_for(MODE, a.BZ, a.Ck, a.Cv, a.T);
    const dim3 grid((a.T + 127) / 128, nsplit, a.BZ);
    const int C1 = MODE == 1 ? a.Ck : a.Cv;
    const long long n0 = (long long)a.BZ * C1 * a.T, n1 = (long long)a.BZ * a.Ck * a.T;
    // one split: the partial IS the result
    float *p0 = (nsplit == 1 || a.keep_parts) ? a.out0 : workspace, *p1 = (nsplit == 1 || a.keep_parts) ? a.out1 : workspace + (size_t)nsplit * n0;
    // split-precision form (round 6): HNO_HM_SPLIT bit MODE (default: all three); 16-byte token quads
    if (X4 && CKT <= 3 && CVT <= 3 && ((hm_split_modes() >> MODE) & 1)) {      // (<= 96 grouped channels: the fragments of wider heads do not fit the registers)
        const size_t lds3 = hm3_lds_bytes<MODE, CKT, CVT>();
        static int attr3 = -1;
        if (lds3 > 48 * 1024 && attr3 != current_device()) {
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)hmha3_kernel<MODE, CKT, CVT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            attr3 = current_device();
        }
        hipLaunchKernelGGL((hmha3_kernel<MODE, CKT, CVT>), grid, dim3(256), lds3, s, a, p0, p1);
    } else
        hipLaunchKernelGGL((hmha2_kernel<MODE, CKT, CVT, X4>), grid, dim3(256), lds, s, a, p0, p1);
    HNO_CHECK_LAUNCH();
    if (nsplit > 1 && !a.keep_parts) {
        const int g0 = (int)((n0 + 255) / 256 < 2048 ? (n0 + 255) / 256 : 2048);
        hipLaunchKernelGGL(hm_sum_partials_kernel, dim3(g0), dim3(256), 0, s, (const float *)p0, a.out0, n0, nsplit);
        if (MODE == 2) {
            const int g1 = (int)((n1 + 255) / 256 < 2048 ? (n1 + 255) / 256 : 2048);
            hipLaunchKernelGGL(hm_sum_partials_kernel, dim3(g1), dim3(256), 0, s, (const float *)p1, a.out1, n1, nsplit);
        }
        HNO_CHECK_LAUNCH();
    }
    return HNO_OK;
}

template <int MODE, int CKT, int CVT>
static int hm_launch(const HmArgs &a, hipStream_t s) {
    constexpr int CMAX = 32 * (CKT > CVT ? CKT : CVT);
    const size_t lds = (size_t)4 * 2 * CMAX * HM_LD * sizeof(float);
    static int attr = -1;
    if (lds > 48 * 1024 && attr != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)hmha_kernel<MODE, CKT, CVT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = current_device();
    }
    const dim3 grid((a.T + 31) / 32, a.BZ);
    hipLaunchKernelGGL((hmha_kernel<MODE, CKT, CVT>), grid, dim3(256), lds, s, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

template <int MODE>
static int hm_dispatch(const HmArgs &a, hipStream_t s, float *workspace = nullptr) {
    const int kt = (a.Ck + 31) / 32, vt = (a.Cv + 31) / 32;
    if (workspace && !(debug_flags() & HNO_DBG_HM_ROUND2) && (a.act == HNO_ACT_NONE || a.act == HNO_ACT_SELU || a.act == HNO_ACT_ELU)) {      // shared-tile kernels (round 4); debug flag HNO_DBG_HM_ROUND2 (1 << 24): the round-2 kernels (A/B)
        if (kt == 1 && vt == 1) return hm2_launch<MODE, 1, 1>(a, workspace, s);
        if (kt <= 2 && vt <= 2) return hm2_launch<MODE, 2, 2>(a, workspace, s);
        if (kt == 3 && vt == 3) return hm2_launch<MODE, 3, 3>(a, workspace, s);
        if (kt <= 4 && vt <= 4) return hm2_launch<MODE, 4, 4>(a, workspace, s);
    }
    if (a.keep_parts) return fail(HNO_ELIMIT, "hno_hmha_*_parts: shape / activation not covered by the shared-tile kernels");
    if (kt == 1 && vt == 1) return hm_launch<MODE, 1, 1>(a, s);
    if (kt == 2 && vt == 2) return hm_launch<MODE, 2, 2>(a, s);
    if (kt == 3 && vt == 3) return hm_launch<MODE, 3, 3>(a, s);
    if (kt == 4 && vt == 4) return hm_launch<MODE, 4, 4>(a, s);
    if (kt <= 2 && vt <= 2) return hm_launch<MODE, 2, 2>(a, s);
    if (kt <= 4 && vt <= 4) return hm_launch<MODE, 4, 4>(a, s);
    return fail(HNO_ELIMIT, "hno_hmha: %d / %d grouped channels per head (max 128)", a.Ck, a.Cv);
}

}  // namespace hno

using namespace hno;

extern "C" int hno_hmha_supported(int Ck, int Cv) { return Ck > 0 && Cv > 0 && Ck <= 128 && Cv <= 128; }

// out (BZ, Cv, T) = V att^T with att = act(alpha Q^T K); q, k (BZ, Ck, T), v (BZ, Cv, T)
// workspace of the shared-tile kernels: partial results of the stream splits, (forward) nsplit x BZ x Cv x T floats,
// (backward) nsplit x BZ x (Ck + Cv) x T; NULL selects the round-2 kernels (no workspace)
extern "C" size_t hno_hmha_workspace_bytes(int BZ, int Ck, int Cv, int T) {
    if (BZ <= 0 || Ck <= 0 || Cv <= 0 || T <= 0) return 0;
    return sizeof(float) * (size_t)hm2_nsplit(BZ, T) * BZ * (size_t)(Ck + Cv) * T;
}

extern "C" int hno_hmha_fwd(const float *q, const float *k, const float *v, float *out, void *workspace, size_t workspace_bytes, int BZ,
                            int Ck, int Cv, int T, float alpha, int act, void *stream) {
    if (workspace && workspace_bytes < hno_hmha_workspace_bytes(BZ, Ck, Cv, T)) workspace = nullptr;
    HNO_REQUIRE(q && k && v && out && BZ > 0 && Ck > 0 && Cv > 0 && T > 0, "hno_hmha_fwd: bad argument");
    HmArgs a = {};
    a.q = q; a.k = k; a.v = v; a.out0 = out; a.BZ = BZ; a.Ck = Ck; a.Cv = Cv; a.T = T; a.alpha = alpha; a.act = act;
    hipStream_t s = (hipStream_t)stream;
    ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (Ck + Cv));
    return hm_dispatch<0>(a, s, (float *)workspace);
}

// gradients of hno_hmha_fwd by recomputation: dq, dk (BZ, Ck, T), dv (BZ, Cv, T) from dout (BZ, Cv, T)
extern "C" int hno_hmha_bwd(const float *q, const float *k, const float *v, const float *dout, float *dq, float *dk, float *dv,
                            void *workspace, size_t workspace_bytes, int BZ, int Ck, int Cv, int T, float alpha, int act, void *stream) {
    if (workspace && workspace_bytes < hno_hmha_workspace_bytes(BZ, Ck, Cv, T)) workspace = nullptr;
    HNO_REQUIRE(q && k && v && dout && dq && dk && dv && BZ > 0 && Ck > 0 && Cv > 0 && T > 0, "hno_hmha_bwd: bad argument");
    HmArgs a = {};
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.BZ = BZ; a.Ck = Ck; a.Cv = Cv; a.T = T; a.alpha = alpha; a.act = act;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (2.0 * Ck + Cv));
        a.out0 = dq;
        const int rc = hm_dispatch<1>(a, s, (float *)workspace);
        if (rc != HNO_OK) return rc;
    }
    ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (2.0 * Ck + 2.0 * Cv));
    a.out0 = dv;
    a.out1 = dk;
    return hm_dispatch<2>(a, s, (float *)workspace);
}

// ---- the stream splits' partial results left unsummed (round 4b): hno_hmha_nsplit(BZ, T) slices of BZ x C x T floats each, which the
// consumer adds in slice order while it reads them anyway (hno_patch_group3_sum) -- hm_sum_partials_kernel was 4 launches of ~7 us per
// attention block and step.  hno_hmha_parts_supported: the shapes / activations the shared-tile kernels serve.
extern "C" int hno_hmha_nsplit(int BZ, int T) { return BZ > 0 && T > 0 ? hm2_nsplit(BZ, T) : 0; }
// slices of hno_hmha_bwd_parts' three outputs (round 6: the backward kernels may take fewer splits than the forward one)
extern "C" int hno_hmha_nsplit_bwd(int BZ, int Ck, int Cv, int T) { return BZ > 0 && Ck > 0 && Cv > 0 && T > 0 ? hm_nsplit_for(1, BZ, Ck, Cv, T) : 0; }

extern "C" int hno_hmha_parts_supported(int Ck, int Cv, int act) {
    return hno_hmha_supported(Ck, Cv) && (act == HNO_ACT_NONE || act == HNO_ACT_SELU || act == HNO_ACT_ELU) && !(debug_flags() & HNO_DBG_HM_ROUND2);
}

extern "C" int hno_hmha_fwd_parts(const float *q, const float *k, const float *v, float *out_parts, int BZ, int Ck, int Cv, int T, float alpha,
                                  int act, void *stream) {
    HNO_REQUIRE(q && k && v && out_parts && BZ > 0 && Ck > 0 && Cv > 0 && T > 0, "hno_hmha_fwd_parts: bad argument");
    if (!hno_hmha_parts_supported(Ck, Cv, act)) return fail(HNO_ELIMIT, "hno_hmha_fwd_parts: not covered");
    HmArgs a = {};
    a.q = q; a.k = k; a.v = v; a.out0 = out_parts; a.BZ = BZ; a.Ck = Ck; a.Cv = Cv; a.T = T; a.alpha = alpha; a.act = act; a.keep_parts = 1;
    hipStream_t s = (hipStream_t)stream;
    ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (Ck + Cv));
    return hm_dispatch<0>(a, s, out_parts);
}

extern "C" int hno_hmha_bwd_parts(const float *q, const float *k, const float *v, const float *dout, float *dq_parts, float *dk_parts,
                                  float *dv_parts, int BZ, int Ck, int Cv, int T, float alpha, int act, void *stream) {
    HNO_REQUIRE(q && k && v && dout && dq_parts && dk_parts && dv_parts && BZ > 0 && Ck > 0 && Cv > 0 && T > 0, "hno_hmha_bwd_parts: bad argument");
    if (!hno_hmha_parts_supported(Ck, Cv, act)) return fail(HNO_ELIMIT, "hno_hmha_bwd_parts: not covered");
    HmArgs a = {};
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.BZ = BZ; a.Ck = Ck; a.Cv = Cv; a.T = T; a.alpha = alpha; a.act = act; a.keep_parts = 1;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (2.0 * Ck + Cv));
        a.out0 = dq_parts;
        const int rc = hm_dispatch<1>(a, s, dq_parts);
        if (rc != HNO_OK) return rc;
    }
    ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (2.0 * Ck + 2.0 * Cv));
    a.out0 = dv_parts;
    a.out1 = dk_parts;
    return hm_dispatch<2>(a, s, dv_parts);
}

// ---- patch grouping (round 4b): grouping3d / ungrouping3d of the reference (nets/hartley_mha.py:473-524) as ONE permutation kernel.
// In the model the q / k / v projections come out of one stacked pointwise convolution as (B, Z Kq + Z Kk + Z Kv, d, h, w); attention wants
// three contiguous (B, Z, K P, T) tensors with P = pd ph pw patch voxels folded into the channels and T = (d / pd)(h / ph)(w / pw) tokens:
//   dst_s[b][(cg P + ((i ph + j) pw + l))][(td nh + th) nw + tw] = src[b][c0_s + cg][td pd + i][th ph + j][tw pw + l]
// As torch ops that was split + 3 x (reshape, permute, copy) forward and the mirror image plus a concatenation backward: ~20 ATen launches
// of ~5 us per attention block and direction (1.7 ms of the 11.8 ms HartleyMHASeg step).  inverse = 1: the other direction (the gradient
// of the grouped tensors back into the stacked layout, and ungrouping3d of the attention output).
struct PgArgs {
    float *full;               // (B, Ctot, d, h, w)
    float *part[3];            // (B, C_s P, T) each
    int cend[3];               // exclusive end channel of each part in the stacked tensor
    int B, Ctot, d, h, w, pd, ph, pw, inverse;
    int nsum;                  // inverse only: every part is `nsum` slices (B C_s P T floats apart) that are added in order
};

__global__ __launch_bounds__(256) void patch_group_kernel(PgArgs a) {
    // blockIdx.y = (b, stacked channel c): everything about the channel is workgroup-uniform; a thread owns tokens and walks the P patch
    // voxels of each (consecutive lanes = consecutive tokens: contiguous in the grouped tensor, pw floats apart in the full one)
    const int nd = a.d / a.pd, nh = a.h / a.ph, nw = a.w / a.pw;
    const int P = a.pd * a.ph * a.pw, T = nd * nh * nw;
    const int b = blockIdx.y / a.Ctot, c = blockIdx.y - b * a.Ctot;
    const int s = c < a.cend[0] ? 0 : (c < a.cend[1] ? 1 : 2);
    const int c0 = s == 0 ? 0 : a.cend[s - 1], Cs = a.cend[s] - c0;
    float *part = a.part[s];
    if (!part && !a.inverse) return;
    float *fullc = a.full + (size_t)blockIdx.y * a.d * a.h * a.w;
    float *partc = part ? part + ((size_t)b * Cs + (c - c0)) * P * T : nullptr;
    // gridDim.z = pd ph: one (i, j) row of the patch per workgroup (few channels: 8 x 48 workgroups of 64 dependent loads per thread were
    // latency-bound at 1.5 TB/s); gridDim.z = 1: a thread walks the whole patch
    const bool zs = gridDim.z > 1;
    const int i0 = zs ? (int)blockIdx.z / a.ph : 0, i1 = zs ? i0 + 1 : a.pd;
    const int j0 = zs ? (int)blockIdx.z - i0 * a.ph : 0, j1 = zs ? j0 + 1 : a.ph;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
        const int tw = t % nw, r = t / nw, th = r % nh, td = r / nh;
        const unsigned base = ((unsigned)(td * a.pd) * a.h + th * a.ph) * a.w + tw * a.pw;
        for (int i = i0; i < i1; ++i)
            for (int j = j0; j < j1; ++j)
                for (int l = 0; l < a.pw; ++l) {
                    const int pi = (i * a.ph + j) * a.pw + l;
                    const unsigned fo = base + ((unsigned)i * a.h + j) * a.w + l;
                    if (a.inverse) {
                        float v = 0.f;
                        if (partc) {
                            v = partc[(size_t)pi * T + t];
                            const size_t slice = (size_t)a.B * Cs * P * T;
                            for (int e = 1; e < a.nsum; ++e) v += partc[(size_t)e * slice + (size_t)pi * T + t];
                        }
                        fullc[fo] = v;
                    } else {
                        partc[(size_t)pi * T + t] = fullc[fo];
                    }
                }
    }
}

// The same permutation with FOUR consecutive tokens of one patch voxel (blockIdx.z) per thread: the grouped side moves as 16-byte
// pieces -- parts -> full: nsum loads per thread added in slice order (the stream-split partial sums of the attention kernels: 8 slices
// of dQ | dK | dV are 72 MB read for 9 MB written in the HartleyMHASeg step, 22.6 us per launch with one token and 64 four-byte loads per
// thread: round 6), four scattered 4-byte stores; full -> parts: four scattered loads, one store.  T % 4 == 0, parts 16-byte aligned.
__global__ __launch_bounds__(256) void patch_group_quad_kernel(PgArgs a) {
    const int nd = a.d / a.pd, nh = a.h / a.ph, nw = a.w / a.pw;
    const int P = a.pd * a.ph * a.pw, T = nd * nh * nw;
    (void)nd;
    const int b = blockIdx.y / a.Ctot, c = blockIdx.y - b * a.Ctot;
    const int s = c < a.cend[0] ? 0 : (c < a.cend[1] ? 1 : 2);
    const int c0 = s == 0 ? 0 : a.cend[s - 1], Cs = a.cend[s] - c0;
    const float *part = a.part[s];
    float *fullc = a.full + (size_t)blockIdx.y * a.d * a.h * a.w;
    const int pi = blockIdx.z, l = pi % a.pw, ij = pi / a.pw, j = ij % a.ph, i = ij / a.ph;
    const unsigned vo = ((unsigned)i * a.h + j) * a.w + l;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    if (!a.inverse) {       // full -> parts: four scattered loads, one 16-byte store
        if (!part) return;
        float vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tk = t + k, tw = tk % nw, r = tk / nw, th = r % nh, td = r / nh;
            vv[k] = fullc[((unsigned)(td * a.pd) * a.h + th * a.ph) * a.w + tw * a.pw + vo];
        }
        *reinterpret_cast<float4 *>(a.part[s] + (((size_t)b * Cs + (c - c0)) * P + pi) * T + t) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        return;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (part) {
        const float *src = part + (((size_t)b * Cs + (c - c0)) * P + pi) * T + t;
        const size_t slice = (size_t)a.B * Cs * P * T;
        v = *reinterpret_cast<const float4 *>(src);
        for (int e = 1; e < a.nsum; ++e) {
            const float4 u = *reinterpret_cast<const float4 *>(src + (size_t)e * slice);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
    }
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tk = t + k, tw = tk % nw, r = tk / nw, th = r % nh, td = r / nh;
        fullc[((unsigned)(td * a.pd) * a.h + th * a.ph) * a.w + tw * a.pw + vo] = vv[k];
    }
}

static int patch_group_launch(float *full, float *p0, float *p1, float *p2, int B, int C0, int C1, int C2, int d, int h, int w, int pd,
                              int ph, int pw, int inverse, int nsum, void *stream);

extern "C" int hno_patch_group3(float *full, float *p0, float *p1, float *p2, int B, int C0, int C1, int C2, int d, int h, int w, int pd,
                                int ph, int pw, int inverse, void *stream) {
    return patch_group_launch(full, p0, p1, p2, B, C0, C1, C2, d, h, w, pd, ph, pw, inverse, 1, stream);
}

// parts -> full with every part given as `nsum` partial slices (hno_hmha_*_parts) that are added in slice order on the way
extern "C" int hno_patch_group3_sum(float *full, const float *p0, const float *p1, const float *p2, int nsum, int B, int C0, int C1, int C2,
                                    int d, int h, int w, int pd, int ph, int pw, void *stream) {
    HNO_REQUIRE(nsum >= 1 && nsum <= 64, "hno_patch_group3_sum: bad slice count");
    return patch_group_launch(full, (float *)p0, (float *)p1, (float *)p2, B, C0, C1, C2, d, h, w, pd, ph, pw, 1, nsum, stream);
}

static int patch_group_launch(float *full, float *p0, float *p1, float *p2, int B, int C0, int C1, int C2, int d, int h, int w, int pd,
                              int ph, int pw, int inverse, int nsum, void *stream) {
    HNO_REQUIRE(full && B > 0 && C0 > 0 && C1 >= 0 && C2 >= 0 && d > 0 && h > 0 && w > 0, "hno_patch_group3: bad argument");
    HNO_REQUIRE(pd > 0 && ph > 0 && pw > 0 && d % pd == 0 && h % ph == 0 && w % pw == 0, "hno_patch_group3: the patch must divide the grid");
    PgArgs a = {};
    a.full = full; a.part[0] = p0; a.part[1] = p1; a.part[2] = p2;
    a.cend[0] = C0; a.cend[1] = C0 + C1; a.cend[2] = C0 + C1 + C2;
    a.B = B; a.Ctot = C0 + C1 + C2; a.d = d; a.h = h; a.w = w; a.pd = pd; a.ph = ph; a.pw = pw; a.inverse = inverse; a.nsum = nsum;
    const int T = (d / pd) * (h / ph) * (w / pw);
    HNO_REQUIRE((long long)B * a.Ctot <= 65535 && (long long)d * h * w < (1ll << 31), "hno_patch_group3: too many channels / voxels");
    static const int quad = getenv("HNO_PG_QUAD") ? atoi(getenv("HNO_PG_QUAD")) : 1;     // A/B aid: 0 = one token per thread
    if (quad && T % 4 == 0 && pd * ph * pw <= 65535 && !(((size_t)p0 | (size_t)p1 | (size_t)p2) & 15)) {
        hipLaunchKernelGGL(patch_group_quad_kernel, dim3((unsigned)((T / 4 + 255) / 256), (unsigned)(B * a.Ctot), (unsigned)(pd * ph * pw)), dim3(256), 0,
                           (hipStream_t)stream, a);
        HNO_CHECK_LAUNCH();
        return HNO_OK;
    }
    int gx = (T + 255) / 256;
    if (gx > 64) gx = 64;
    static const int split_mode = getenv("HNO_PG_SPLIT") ? atoi(getenv("HNO_PG_SPLIT")) : 1;     // 0: never, 1: below 1 024 workgroups, 2: always
    const bool split = pd * ph > 1 && pd * ph <= 64 && (split_mode == 2 || (split_mode == 1 && (long long)gx * B * a.Ctot < 1024));
    hipLaunchKernelGGL(patch_group_kernel, dim3((unsigned)gx, (unsigned)(B * a.Ctot), split ? (unsigned)(pd * ph) : 1u), dim3(256), 0,
                       (hipStream_t)stream, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
