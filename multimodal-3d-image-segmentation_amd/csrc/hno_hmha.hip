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
#include "hno_common.h"

namespace hno {

typedef float f32x16h __attribute__((ext_vector_type(16)));

struct HmArgs {
    const float *q, *k, *v, *dout;   // (BZ, Ck, T), (BZ, Ck, T), (BZ, Cv, T), (BZ, Cv, T)
    float *out0, *out1;              // MODE 0: out (BZ, Cv, T); MODE 1: dQ (BZ, Ck, T); MODE 2: dV (out0), dK (out1)
    int BZ, Ck, Cv, T;
    float alpha;
    int act;
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
static int hm_dispatch(const HmArgs &a, hipStream_t s) {
    const int kt = (a.Ck + 31) / 32, vt = (a.Cv + 31) / 32;
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
extern "C" int hno_hmha_fwd(const float *q, const float *k, const float *v, float *out, int BZ, int Ck, int Cv, int T, float alpha,
                            int act, void *stream) {
    HNO_REQUIRE(q && k && v && out && BZ > 0 && Ck > 0 && Cv > 0 && T > 0, "hno_hmha_fwd: bad argument");
    HmArgs a = {};
    a.q = q; a.k = k; a.v = v; a.out0 = out; a.BZ = BZ; a.Ck = Ck; a.Cv = Cv; a.T = T; a.alpha = alpha; a.act = act;
    hipStream_t s = (hipStream_t)stream;
    ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (Ck + Cv));
    return hm_dispatch<0>(a, s);
}

// gradients of hno_hmha_fwd by recomputation: dq, dk (BZ, Ck, T), dv (BZ, Cv, T) from dout (BZ, Cv, T)
extern "C" int hno_hmha_bwd(const float *q, const float *k, const float *v, const float *dout, float *dq, float *dk, float *dv, int BZ,
                            int Ck, int Cv, int T, float alpha, int act, void *stream) {
    HNO_REQUIRE(q && k && v && dout && dq && dk && dv && BZ > 0 && Ck > 0 && Cv > 0 && T > 0, "hno_hmha_bwd: bad argument");
    HmArgs a = {};
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.BZ = BZ; a.Ck = Ck; a.Cv = Cv; a.T = T; a.alpha = alpha; a.act = act;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (2.0 * Ck + Cv));
        a.out0 = dq;
        const int rc = hm_dispatch<1>(a, s);
        if (rc != HNO_OK) return rc;
    }
    ProfScope _ps(KID_HMHA, s, 2.0 * BZ * (double)T * T * (2.0 * Ck + 2.0 * Cv));
    a.out0 = dv;
    a.out1 = dk;
    return hm_dispatch<2>(a, s);
}
