// The deep-supervision convolution over the channel concat of T equally wide tensors (reference nets/architectures.py:341-343, :196-199:
// torch.cat(block outputs, dim=1) -> ConvNormAct(k = 1)) without the concat AND without one launch per tensor:
//     out[b, k, v] = bias[k] + sum_t sum_c W[t][k][c] x_t[b, c, v]
// Rounds 2-6 ran it as T pointwise launches + T - 1 adds forward and T pointwise-backward launches + T slab reductions backward
// (nets/deep_supervision.py): for HartleyMHASeg's 17 legs of 12 channels that was 66 launches and 0.52 ms of a 10.4 ms step, every one of
// them a few MB at the launch floor.  Here: ONE forward launch (a thread per voxel walks the legs; the T x K x C weights sit in LDS,
// [t][c][k] so that the K weights of an input element are one read) and ONE backward launch (gridDim.y = leg: input gradient
// gx_t = W_t^T g and the leg's K x C weight-gradient partials, one slab row per (chunk, sample)) + the usual fixed-order slab reduction.
// fp32 FMA chains (VALU: 2 K flops per loaded element, far below the load rate), channel-padded activations (row stride ld >= V; the
// padding is neither read nor written).  Built for 2 ... 5 output channels (classes), 8 / 12 / 16 / 24 channels per leg and up to 32
// legs (hno_pwmulti_supported); other shapes keep the per-leg path.
#include "hno_common.h"

namespace hno {

#define PWM_MAXT 32
#define PWM_MAXK 8

struct PwMultiArgs {
    const float *x[PWM_MAXT];
    float *gx[PWM_MAXT];
    const float *W, *bias, *g;       // W: [T][K][C]
    float *out, *slab;
    int T, C, K, B, nslab_cols, vec4;
    unsigned V, ld;
};

__global__ __launch_bounds__(256) void pwmulti_fwd_kernel(PwMultiArgs a) {
    extern __shared__ float wl[];          // [t][c][PWM_MAXK] (k padded to 8: two 16-byte reads per input element), then the bias
    const int T = a.T, C = a.C, K = a.K;
    for (int i = threadIdx.x; i < T * C * PWM_MAXK; i += 256) {
        const int k = i & (PWM_MAXK - 1), tc = i >> 3, t = tc / C, c = tc - t * C;
        wl[i] = k < K ? a.W[((size_t)t * K + k) * C + c] : 0.f;
    }
    float *bl = wl + T * C * PWM_MAXK;
    if (threadIdx.x < PWM_MAXK) bl[threadIdx.x] = (a.bias && (int)threadIdx.x < K) ? a.bias[threadIdx.x] : 0.f;
    __syncthreads();
    const int b = blockIdx.y;
    // four consecutive voxels per thread (16-byte loads: a wave takes 1 KB of every one of the T C row streams per step; one voxel per
    // thread -- 256-byte pieces of 204 streams -- ran at 2.1 TB/s); V % 4 == 0 and 16-byte aligned rows (host check), else VEC = 1
    if (a.vec4) {
        for (unsigned v = (blockIdx.x * 256 + threadIdx.x) * 4; v < a.V; v += gridDim.x * 1024) {
            float4 acc[PWM_MAXK];
#pragma unroll
            for (int k = 0; k < PWM_MAXK; ++k) acc[k] = make_float4(bl[k], bl[k], bl[k], bl[k]);
            for (int t = 0; t < T; ++t) {
                const float *xt = a.x[t] + (size_t)b * C * a.ld + v;
                const float *wt = wl + (size_t)t * C * PWM_MAXK;
#pragma unroll 4
                for (int c = 0; c < C; ++c) {
                    const float4 xv = *reinterpret_cast<const float4 *>(xt + (size_t)c * a.ld);
                    const float4 w0 = *reinterpret_cast<const float4 *>(wt + c * PWM_MAXK), w1 = *reinterpret_cast<const float4 *>(wt + c * PWM_MAXK + 4);
                    const float wk[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int k = 0; k < PWM_MAXK; ++k)
                        if (k < 4 || K > 4) {
                            acc[k].x = fmaf(wk[k], xv.x, acc[k].x); acc[k].y = fmaf(wk[k], xv.y, acc[k].y);
                            acc[k].z = fmaf(wk[k], xv.z, acc[k].z); acc[k].w = fmaf(wk[k], xv.w, acc[k].w);
                        }
                }
            }
            float *o = a.out + (size_t)b * K * a.ld + v;
#pragma unroll
            for (int k = 0; k < PWM_MAXK; ++k)
                if (k < K) *reinterpret_cast<float4 *>(o + (size_t)k * a.ld) = acc[k];
        }
        return;
    }
    for (unsigned v = blockIdx.x * 256 + threadIdx.x; v < a.V; v += gridDim.x * 256) {
        float acc[PWM_MAXK];
#pragma unroll
        for (int k = 0; k < PWM_MAXK; ++k) acc[k] = bl[k];
        for (int t = 0; t < T; ++t) {
            const float *xt = a.x[t] + (size_t)b * C * a.ld + v;
            const float *wt = wl + (size_t)t * C * PWM_MAXK;
#pragma unroll 4
            for (int c = 0; c < C; ++c) {
                const float xv = xt[(size_t)c * a.ld];
                const float4 w0 = *reinterpret_cast<const float4 *>(wt + c * PWM_MAXK), w1 = *reinterpret_cast<const float4 *>(wt + c * PWM_MAXK + 4);
                acc[0] = fmaf(w0.x, xv, acc[0]); acc[1] = fmaf(w0.y, xv, acc[1]); acc[2] = fmaf(w0.z, xv, acc[2]); acc[3] = fmaf(w0.w, xv, acc[3]);
                if (K > 4) {      // (uniform)
                    acc[4] = fmaf(w1.x, xv, acc[4]); acc[5] = fmaf(w1.y, xv, acc[5]); acc[6] = fmaf(w1.z, xv, acc[6]); acc[7] = fmaf(w1.w, xv, acc[7]);
                }
            }
        }
        float *o = a.out + (size_t)b * K * a.ld + v;
#pragma unroll
        for (int k = 0; k < PWM_MAXK; ++k)
            if (k < K) o[(size_t)k * a.ld] = acc[k];
    }
}

// backward of one leg per blockIdx.y (K, C at compile time: the K x C weight-gradient sums of a thread live in registers).
// slab row (chunk, sample) = [T][K][C] weight-gradient partials + K bias-gradient partials (written by the workgroups of leg 0).
template <int K, int C, int VEC>
__global__ __launch_bounds__(256) void pwmulti_bwd_kernel(PwMultiArgs a) {
    __shared__ float wl[K * C];            // W_t [k][c]
    __shared__ float red[4][K * C + K];
    const int t = blockIdx.y, b = blockIdx.z;
    for (int i = threadIdx.x; i < K * C; i += 256) wl[i] = a.W[(size_t)t * K * C + i];
    __syncthreads();
    float w[K][C];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < C; ++c) w[k][c] = wl[k * C + c];
    float dw[K][C], db[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        db[k] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) dw[k][c] = 0.f;
    }
    const float *xt = a.x[t] + (size_t)b * C * a.ld;
    float *gxt = a.gx[t] ? a.gx[t] + (size_t)b * C * a.ld : nullptr;
    const float *gb = a.g + (size_t)b * K * a.ld;
    // this workgroup's chunk of voxels: contiguous (a multiple of 4 voxels), so that consecutive workgroups stream consecutive lines;
    // VEC = 4: four consecutive voxels per thread, 16-byte loads and stores (see the forward kernel)
    const unsigned per = ((a.V + gridDim.x - 1) / gridDim.x + 3) & ~3u, lo = blockIdx.x * per, hi = lo + per < a.V ? lo + per : a.V;
    typedef float vec __attribute__((ext_vector_type(VEC)));
    for (unsigned v = lo + threadIdx.x * VEC; v < hi; v += 256 * VEC) {
        vec g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = *reinterpret_cast<const vec *>(gb + (size_t)k * a.ld + v);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) db[k] += g[k][e];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const vec xv = *reinterpret_cast<const vec *>(xt + (size_t)c * a.ld + v);
            vec s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    s[e] = fmaf(w[k][c], g[k][e], s[e]);
                    dw[k][c] = fmaf(g[k][e], xv[e], dw[k][c]);
                }
            }
            if (gxt) *reinterpret_cast<vec *>(gxt + (size_t)c * a.ld + v) = s;
        }
    }
    // K C + K sums: over the 64 lanes of a wave (butterfly), over the 4 waves through LDS, one slab row per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float s = dw[k][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red[wave][k * C + c] = s;
        }
        float s = db[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave][K * C + k] = s;
    }
    __syncthreads();
    float *row = a.slab + ((size_t)b * gridDim.x + blockIdx.x) * a.nslab_cols;
    for (int i = threadIdx.x; i < K * C; i += 256) row[(size_t)t * K * C + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    if (t == 0 && (int)threadIdx.x < K) {
        const int i = K * C + threadIdx.x;
        row[(size_t)a.T * K * C + threadIdx.x] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

static int pwm_chunks(int T, int B, long long V) {
    long long n = 2048 / ((long long)T * B);       // ~2 048 workgroups in the backward launch
    const long long cap = (V + 1023) / 1024;       // at least four voxels per thread
    if (n > cap) n = cap;
    if (n > 256) n = 256;
    return n < 1 ? 1 : (int)n;
}

}  // namespace hno

using namespace hno;

extern "C" int hno_pwmulti_supported(int T, int C, int K) {
    return T >= 1 && T <= PWM_MAXT && (K >= 2 && K <= 5) && (C == 8 || C == 12 || C == 16 || C == 24);
}

extern "C" size_t hno_pwmulti_bwd_workspace_bytes(int T, int C, int K, int B, long long V) {
    if (T < 1 || C < 1 || K < 1 || B < 1 || V < 1) return 0;
    return sizeof(float) * (size_t)pwm_chunks(T, B, V) * B * ((size_t)T * K * C + K);
}

static int pwm_fill(PwMultiArgs &a, const void *const *x, void *const *gx, int T, int C, int K, int B, long long V, long long ld, const char *who) {
    HNO_REQUIRE(x && T >= 1 && T <= PWM_MAXT && C >= 1 && K >= 1 && K <= PWM_MAXK && B >= 1 && V >= 1 && ld >= V, "%s: bad argument", who);
    if (ld >= (1ll << 31)) return fail(HNO_ELIMIT, "%s: %lld voxels per channel exceed the 32-bit voxel index", who, ld);
    for (int t = 0; t < PWM_MAXT; ++t) {
        a.x[t] = t < T ? (const float *)x[t] : nullptr;
        a.gx[t] = (gx && t < T) ? (float *)gx[t] : nullptr;
        HNO_REQUIRE(t >= T || a.x[t], "%s: input %d missing", who, t);
    }
    a.T = T; a.C = C; a.K = K; a.B = B; a.V = (unsigned)V; a.ld = (unsigned)ld;
    bool al = V % 4 == 0 && ld % 4 == 0;
    for (int t = 0; t < T; ++t) al = al && !((size_t)a.x[t] & 15) && !((size_t)a.gx[t] & 15);
    a.vec4 = al ? 1 : 0;
    return HNO_OK;
}

// out (B, K, ld) = bias + sum_t W[t] x_t;  x: HOST array of T device pointers to (B, C, ld) tensors;  W: [T][K][C]
extern "C" int hno_pwmulti_fwd(const void *const *x, int T, int C, const float *W, const float *bias, float *out, int B, int K, long long V,
                               long long ld, void *stream) {
    PwMultiArgs a = {};
    const int rc = pwm_fill(a, x, nullptr, T, C, K, B, V, ld, "hno_pwmulti_fwd");
    if (rc != HNO_OK) return rc;
    HNO_REQUIRE(W && out, "hno_pwmulti_fwd: bad argument");
    a.W = W; a.bias = bias; a.out = out;
    if ((size_t)out & 15) a.vec4 = 0;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(float) * ((size_t)T * C * PWM_MAXK + PWM_MAXK);
    if (lds > 64 * 1024) return fail(HNO_ELIMIT, "hno_pwmulti_fwd: %d x %d weights exceed the kernel's LDS image", T, C);
    long long gx_ = (V + (a.vec4 ? 1023 : 255)) / (a.vec4 ? 1024 : 256);
    if (gx_ > 2048) gx_ = 2048;
    ProfScope ps(KID_PWCONV_FWD, s, 4.0 * B * (double)V * ((double)T * C + K));
    hipLaunchKernelGGL(pwmulti_fwd_kernel, dim3((unsigned)gx_, B), dim3(256), lds, s, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

template <int K, int C>
static void pwm_bwd_launch(const PwMultiArgs &a, dim3 grid, hipStream_t s) {
    if (a.vec4) hipLaunchKernelGGL((pwmulti_bwd_kernel<K, C, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pwmulti_bwd_kernel<K, C, 1>), grid, dim3(256), 0, s, a);
}

// gx[t] (B, C, ld) = W[t]^T g (NULL entries / NULL array: not wanted), dW [T][K][C], dbias [K] (NULL: none) from g (B, K, ld)
extern "C" int hno_pwmulti_bwd(const float *g, const void *const *x, void *const *gx, int T, int C, const float *W, float *dW, float *dbias,
                               void *workspace, size_t workspace_bytes, int B, int K, long long V, long long ld, void *stream) {
    PwMultiArgs a = {};
    const int rc = pwm_fill(a, x, gx, T, C, K, B, V, ld, "hno_pwmulti_bwd");
    if (rc != HNO_OK) return rc;
    HNO_REQUIRE(g && W && dW && workspace, "hno_pwmulti_bwd: bad argument");
    if (!hno_pwmulti_supported(T, C, K)) return fail(HNO_ELIMIT, "hno_pwmulti_bwd: %d legs of %d -> %d channels are not built", T, C, K);
    HNO_REQUIRE(workspace_bytes >= hno_pwmulti_bwd_workspace_bytes(T, C, K, B, V), "hno_pwmulti_bwd: workspace too small");
    a.g = g; a.W = W; a.slab = (float *)workspace; a.nslab_cols = T * K * C + K;
    if ((size_t)g & 15) a.vec4 = 0;
    hipStream_t s = (hipStream_t)stream;
    const int nch = pwm_chunks(T, B, V);
    const dim3 grid(nch, T, B);
    {
        ProfScope ps(KID_PWCONV_BWD, s, 4.0 * B * (double)V * ((double)T * 2 * C + (double)T * K));
#define PWM_CASE(Kv, Cv) if (K == Kv && C == Cv) pwm_bwd_launch<Kv, Cv>(a, grid, s);
        PWM_CASE(2, 8) PWM_CASE(2, 12) PWM_CASE(2, 16) PWM_CASE(2, 24) PWM_CASE(3, 8) PWM_CASE(3, 12) PWM_CASE(3, 16) PWM_CASE(3, 24)
        PWM_CASE(4, 8) PWM_CASE(4, 12) PWM_CASE(4, 16) PWM_CASE(4, 24) PWM_CASE(5, 8) PWM_CASE(5, 12) PWM_CASE(5, 16) PWM_CASE(5, 24)
#undef PWM_CASE
        HNO_CHECK_LAUNCH();
    }
    // (the slab belongs to this call's workspace tensor: reduced now, in the fixed order of every other weight gradient)
    return reduce_partials_launch(a.slab, nch * B, a.nslab_cols, dW, T * K * C, dbias, s, 0, 0, false);
}
