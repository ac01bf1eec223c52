// Batched fp32 GEMM on the matrix cores: C[b] = alpha * op(A[b]) * op(B[b]), row-major operands.
//
// Used by the Hartley multi-head attention (reference nets/hartley_mha.py:196-201:
// att = einsum('bzcq,bzck->bzqk') / sqrt(C), out = einsum('bzqk,bzck->bzcq')) and its backward.
// 64x64 output tile per 256-thread workgroup, K-steps of 16 staged through LDS, each wave owns a
// 32x32 sub-tile on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).
#include "hno_common.h"

namespace hno {

typedef float f32x16b __attribute__((ext_vector_type(16)));

struct BmmArgs {
    const float *A, *B;
    float *C;
    int M, N, K, lda, ldb, ldc;
    long long sA, sB, sC;  // batch strides (elements)
    int transA, transB;
    float alpha;
};

#define BMM_K 16

struct __attribute__((packed, aligned(4))) bmm_f4u { float x, y, z, w; };   // 16-byte load at 4-byte alignment
typedef float bmm_f4 __attribute__((ext_vector_type(4)));

// One operand tile of TM rows (m or n) x BMM_K, fetched into registers as 16-byte vectors and later written to
// LDS as T[k][m].  KMAJOR: stored k-major (m contiguous: transA for A, !transB for B); else m-major (k contiguous).
template <int TM>
struct BmmStage {
    static constexpr int NV = TM * BMM_K / 4 / 256;   // float4 per thread
    bmm_f4 v[NV];
    template <bool KMAJOR>
    __device__ __forceinline__ void fetch(const float *base, int ld, int m0, int k0, int Mdim, int Kdim, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = tid + 256 * j;
            int m, k;
            if (KMAJOR) {
                k = e / (TM / 4);
                m = (e - k * (TM / 4)) * 4;
            } else {
                m = e / (BMM_K / 4);
                k = (e - m * (BMM_K / 4)) * 4;
            }
            const int gm = m0 + m, gk = k0 + k;
            const float *p = KMAJOR ? base + (size_t)gk * ld + gm : base + (size_t)gm * ld + gk;
            const bool full = KMAJOR ? (gk < Kdim && gm + 3 < Mdim) : (gm < Mdim && gk + 3 < Kdim);
            if (full) {
                const bmm_f4u t = *reinterpret_cast<const bmm_f4u *>(p);
                v[j] = bmm_f4{t.x, t.y, t.z, t.w};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = KMAJOR ? (gk < Kdim && gm + i < Mdim) : (gm < Mdim && gk + i < Kdim);
                    v[j][i] = ok ? p[i] : 0.f;   // both layouts: the 4 elements are contiguous in memory
                }
            }
        }
    }
    template <bool KMAJOR>
    __device__ __forceinline__ void store(float *T, int ldt, int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = tid + 256 * j;
            if (KMAJOR) {
                const int k = e / (TM / 4), m = (e - k * (TM / 4)) * 4;
                *reinterpret_cast<bmm_f4 *>(T + k * ldt + m) = v[j];
            } else {
                const int m = e / (BMM_K / 4), k = (e - m * (BMM_K / 4)) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) T[(k + i) * ldt + m] = v[j][i];
            }
        }
    }
};

// (64 WM) x (64 WN) output tile per 256-thread workgroup, 2 x 2 waves, each wave WM x WN MFMA tiles of 32 x 32.
// K-steps of 16 double-buffered in LDS; the next step's operands are fetched into registers (16-byte loads along
// the contiguous dimension of each operand) while the current step's 8 WM WN MFMAs per wave run.
template <int WM, int WN, bool AK, bool BK>   // AK / BK: A / B stored k-major
__global__ __launch_bounds__(256) void bmm_kernel(BmmArgs a) {
    constexpr int BM = 64 * WM, BN = 64 * WN, LDA = BM + 4, LDB = BN + 4;
    __shared__ float As[2][BMM_K * LDA], Bs[2][BMM_K * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float *Ab = a.A + (size_t)blockIdx.z * a.sA, *Bb = a.B + (size_t)blockIdx.z * a.sB;
    float *Cb = a.C + (size_t)blockIdx.z * a.sC;
    const int wm = (wave >> 1) * 32 * WM, wn = (wave & 1) * 32 * WN;
    f32x16b acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    BmmStage<BM> sa;
    BmmStage<BN> sb;
    sa.template fetch<AK>(Ab, a.lda, m0, 0, a.M, a.K, tid);
    sb.template fetch<BK>(Bb, a.ldb, n0, 0, a.N, a.K, tid);
    sa.template store<AK>(As[0], LDA, tid);
    sb.template store<BK>(Bs[0], LDB, tid);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < a.K; k0 += BMM_K, buf ^= 1) {
        const bool more = k0 + BMM_K < a.K;
        if (more) {
            sa.template fetch<AK>(Ab, a.lda, m0, k0 + BMM_K, a.M, a.K, tid);
            sb.template fetch<BK>(Bb, a.ldb, n0, k0 + BMM_K, a.N, a.K, tid);
        }
        const float *Ac = As[buf] + (lane >> 5) * LDA + wm + (lane & 31);
        const float *Bc = Bs[buf] + (lane >> 5) * LDB + wn + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BMM_K; kk += 2) {
            float av[WM], bv[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) av[i] = Ac[kk * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < WN; ++j) bv[j] = Bc[kk * LDB + 32 * j];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            sa.template store<AK>(As[buf ^ 1], LDA, tid);
            sb.template store<BK>(Bs[buf ^ 1], LDB, tid);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < a.M && col < a.N) Cb[(size_t)row * a.ldc + col] = a.alpha * acc[i][j][r];
            }
        }
}

template <int WM, int WN>
static void bmm_dispatch(const BmmArgs &a, int batch, hipStream_t s) {
    const dim3 g(ceil_div(a.N, 64 * WN), ceil_div(a.M, 64 * WM), batch), b(256);
    const bool ak = a.transA != 0, bk = a.transB == 0;
    if (ak && bk) hipLaunchKernelGGL((bmm_kernel<WM, WN, true, true>), g, b, 0, s, a);
    else if (ak) hipLaunchKernelGGL((bmm_kernel<WM, WN, true, false>), g, b, 0, s, a);
    else if (bk) hipLaunchKernelGGL((bmm_kernel<WM, WN, false, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((bmm_kernel<WM, WN, false, false>), g, b, 0, s, a);
}

}  // namespace hno

using namespace hno;

extern "C" int hno_bmm(const float *A, const float *B, float *C, int batch, int M, int N, int K, int transA, int transB,
                       float alpha, void *stream) {
    HNO_REQUIRE(A && B && C && batch > 0 && M > 0 && N > 0 && K > 0, "hno_bmm: bad argument");
    if (batch > 65535) return fail(HNO_ELIMIT, "hno_bmm: batch %d exceeds 65535", batch);
    BmmArgs a;
    a.A = A; a.B = B; a.C = C;
    a.M = M; a.N = N; a.K = K;
    a.lda = transA ? M : K;
    a.ldb = transB ? K : N;
    a.ldc = N;
    a.sA = (long long)M * K; a.sB = (long long)K * N; a.sC = (long long)M * N;
    a.transA = transA; a.transB = transB; a.alpha = alpha;
    ProfScope _ps0(KID_BMM, (hipStream_t)stream, 4.0 * batch * ((double)M * K + (double)K * N + (double)M * N));
    // 128 x 128 tiles when they still give at least ~one workgroup per CU, else 64 x 64 (e.g. the 96 x 1960 AV product)
    const long long big = (long long)ceil_div(N, 128) * ceil_div(M, 128) * batch;
    if (big >= 192 && M > 64 && N > 64) bmm_dispatch<2, 2>(a, batch, (hipStream_t)stream);
    else bmm_dispatch<1, 1>(a, batch, (hipStream_t)stream);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
