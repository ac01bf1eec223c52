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

#define BMM_T 64
#define BMM_K 16
#define BMM_LD 72

__global__ __launch_bounds__(256) void bmm_kernel(BmmArgs a) {
    __shared__ float As[BMM_K * BMM_LD], Bs[BMM_K * BMM_LD];  // [k][m] and [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BMM_T, n0 = blockIdx.x * BMM_T;
    const float *Ab = a.A + (size_t)blockIdx.z * a.sA, *Bb = a.B + (size_t)blockIdx.z * a.sB;
    float *Cb = a.C + (size_t)blockIdx.z * a.sC;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16b acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += BMM_K) {
        // stage op(A)[m0..m0+63][k0..k0+15] as As[k][m] and op(B)[k][n0..] as Bs[k][n]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            int m, k;
            if (a.transA) {  // A stored K x M: contiguous along m
                m = e & 63;
                k = e >> 6;
            } else {         // A stored M x K: contiguous along k
                k = e & 15;
                m = e >> 4;
            }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < a.M && gk < a.K) v = a.transA ? Ab[(size_t)gk * a.lda + gm] : Ab[(size_t)gm * a.lda + gk];
            As[k * BMM_LD + m] = v;
            int n;
            if (a.transB) {  // B stored N x K: contiguous along k
                k = e & 15;
                n = e >> 4;
            } else {         // B stored K x N: contiguous along n
                n = e & 63;
                k = e >> 6;
            }
            const int gn = n0 + n;
            const int gk2 = k0 + k;
            v = 0.f;
            if (gn < a.N && gk2 < a.K) v = a.transB ? Bb[(size_t)gn * a.ldb + gk2] : Bb[(size_t)gk2 * a.ldb + gn];
            Bs[k * BMM_LD + n] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BMM_K; kk += 2) {
            const float av = As[(kk + (lane >> 5)) * BMM_LD + wm + (lane & 31)];
            const float bv = Bs[(kk + (lane >> 5)) * BMM_LD + wn + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < a.M && col < a.N) Cb[(size_t)row * a.ldc + col] = a.alpha * acc[r];
    }
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
    hipLaunchKernelGGL(bmm_kernel, dim3(ceil_div(N, BMM_T), ceil_div(M, BMM_T), batch), dim3(256), 0, (hipStream_t)stream, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
