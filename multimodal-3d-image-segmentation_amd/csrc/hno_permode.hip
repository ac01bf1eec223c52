// Per-mode ('individual') spectral weights: every kept frequency has its own Cout x Cin matrix.
//
// Reference: FourierOperator 'oidhw,bidhw->bodhw' with complex weights (nets/fourier_operator.py:174-191)
// and HartleyOperator.hartley_conv (nets/hartley_operator.py:302-317):
//     y(k) = 1/2 [ W(k) (x(k) + x(-k)) + W(-k) (x(k) - x(-k)) ],   x(-k) = roll(flip(x), 1)
// The FNO configuration carries 15.9 M such weights, so the op is bound by streaming W once
// (arithmetic intensity 2*B flop per weight): one lane per mode, weights read exactly once and
// reused for every batch element, all accesses coalesced along the mode axis.
#include "hno_common.h"

namespace hno {

struct PmArgs {
    const float *x, *xr;    // Hartley: x and its frequency-reversed copy xr (B, Ci, M); Fourier: x = [re | im] planes (B, 2, Ci, M)
    const float *w, *wi;    // Hartley: w (Co, Ci, M); Fourier: w = real part, wi = imaginary part
    const float *g;         // backward: dL/dy, same layout as y
    float *y, *gx, *gxr, *dw, *dwi;
    int B, Ci, Co, M;
    int d0, d1, d2;          // Hartley mode grid (2m0, 2m1, 2m2) for the index reversal
    int fourier;
};

__device__ __forceinline__ int rev_index(int k, int d0, int d1, int d2) {
    const int k2 = k % d2, k1 = (k / d2) % d1, k0 = k / (d2 * d1);
    const int r0 = k0 ? d0 - k0 : 0, r1 = k1 ? d1 - k1 : 0, r2 = k2 ? d2 - k2 : 0;
    return (r0 * d1 + r1) * d2 + r2;
}

template <int BMAX>
__global__ __launch_bounds__(256) void permode_fwd_kernel(PmArgs a) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.M) return;
    const int o = blockIdx.y;
    const size_t M = a.M;
    if (a.fourier) {
        float yr[BMAX], yi[BMAX];
#pragma unroll
        for (int b = 0; b < BMAX; ++b) yr[b] = yi[b] = 0.f;
        for (int i = 0; i < a.Ci; ++i) {
            const float wr = a.w[((size_t)o * a.Ci + i) * M + k], wim = a.wi[((size_t)o * a.Ci + i) * M + k];
#pragma unroll
            for (int b = 0; b < BMAX; ++b)
                if (b < a.B) {
                    const float xr = a.x[(((size_t)b * 2 + 0) * a.Ci + i) * M + k], xi = a.x[(((size_t)b * 2 + 1) * a.Ci + i) * M + k];
                    yr[b] += wr * xr - wim * xi;
                    yi[b] += wr * xi + wim * xr;
                }
        }
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
            if (b < a.B) {
                a.y[(((size_t)b * 2 + 0) * a.Co + o) * M + k] = yr[b];
                a.y[(((size_t)b * 2 + 1) * a.Co + o) * M + k] = yi[b];
            }
    } else {
        // the weight's own reversal always lives on the (2m)^3 grid; x(-k) comes from the caller (xr)
        const int rk = rev_index(k, a.d0, a.d1, a.d2);
        float acc[BMAX];
#pragma unroll
        for (int b = 0; b < BMAX; ++b) acc[b] = 0.f;
        for (int i = 0; i < a.Ci; ++i) {
            const float wk = a.w[((size_t)o * a.Ci + i) * M + k], wr = a.w[((size_t)o * a.Ci + i) * M + rk];
#pragma unroll
            for (int b = 0; b < BMAX; ++b)
                if (b < a.B) {
                    const float xk = a.x[((size_t)b * a.Ci + i) * M + k], xrv = a.xr[((size_t)b * a.Ci + i) * M + k];
                    acc[b] += wk * (xk + xrv) + wr * (xk - xrv);
                }
        }
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
            if (b < a.B) a.y[((size_t)b * a.Co + o) * M + k] = 0.5f * acc[b];
    }
}

// input gradients: one lane per (input channel i, mode k)
template <int BMAX>
__global__ __launch_bounds__(256) void permode_dgrad_kernel(PmArgs a) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.M) return;
    const int i = blockIdx.y;
    const size_t M = a.M;
    if (a.fourier) {
        float gr[BMAX], gi[BMAX];
#pragma unroll
        for (int b = 0; b < BMAX; ++b) gr[b] = gi[b] = 0.f;
        for (int o = 0; o < a.Co; ++o) {
            const float wr = a.w[((size_t)o * a.Ci + i) * M + k], wim = a.wi[((size_t)o * a.Ci + i) * M + k];
#pragma unroll
            for (int b = 0; b < BMAX; ++b)
                if (b < a.B) {
                    const float yr = a.g[(((size_t)b * 2 + 0) * a.Co + o) * M + k], yi = a.g[(((size_t)b * 2 + 1) * a.Co + o) * M + k];
                    gr[b] += wr * yr + wim * yi;     // conj(W)^T g
                    gi[b] += wr * yi - wim * yr;
                }
        }
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
            if (b < a.B) {
                a.gx[(((size_t)b * 2 + 0) * a.Ci + i) * M + k] = gr[b];
                a.gx[(((size_t)b * 2 + 1) * a.Ci + i) * M + k] = gi[b];
            }
    } else {
        const int rk = rev_index(k, a.d0, a.d1, a.d2);
        float s[BMAX], d[BMAX];
#pragma unroll
        for (int b = 0; b < BMAX; ++b) s[b] = d[b] = 0.f;
        for (int o = 0; o < a.Co; ++o) {
            const float wk = a.w[((size_t)o * a.Ci + i) * M + k], wr = a.w[((size_t)o * a.Ci + i) * M + rk];
#pragma unroll
            for (int b = 0; b < BMAX; ++b)
                if (b < a.B) {
                    const float gv = a.g[((size_t)b * a.Co + o) * M + k];
                    s[b] += (wk + wr) * gv;   // dy/dx(k)
                    d[b] += (wk - wr) * gv;   // dy/dxr(k)
                }
        }
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
            if (b < a.B) {
                a.gx[((size_t)b * a.Ci + i) * M + k] = 0.5f * s[b];
                a.gxr[((size_t)b * a.Ci + i) * M + k] = 0.5f * d[b];
            }
    }
}

// weight gradients: one lane per (o, i, k); no cross-mode reduction exists
template <int BMAX>
__global__ __launch_bounds__(256) void permode_wgrad_kernel(PmArgs a) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.M) return;
    const int i = blockIdx.y % a.Ci, o = blockIdx.y / a.Ci;
    const size_t M = a.M;
    if (a.fourier) {
        float dr = 0.f, di = 0.f;
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
            if (b < a.B) {
                const float xr = a.x[(((size_t)b * 2 + 0) * a.Ci + i) * M + k], xi = a.x[(((size_t)b * 2 + 1) * a.Ci + i) * M + k];
                const float yr = a.g[(((size_t)b * 2 + 0) * a.Co + o) * M + k], yi = a.g[(((size_t)b * 2 + 1) * a.Co + o) * M + k];
                dr += yr * xr + yi * xi;
                di += yi * xr - yr * xi;
            }
        a.dw[((size_t)o * a.Ci + i) * M + k] = dr;
        a.dwi[((size_t)o * a.Ci + i) * M + k] = di;
    } else {
        // dW(k) = 1/2 [ sum_b g(k) (x(k) + xr(k)) + g(rk) (x(rk) - xr(rk)) ]  (W(k) is used at k and at rk)
        const int rk = rev_index(k, a.d0, a.d1, a.d2);
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
            if (b < a.B) {
                const size_t xo = ((size_t)b * a.Ci + i) * M, go = ((size_t)b * a.Co + o) * M;
                acc += a.g[go + k] * (a.x[xo + k] + a.xr[xo + k]) + a.g[go + rk] * (a.x[xo + rk] - a.xr[xo + rk]);
            }
        a.dw[((size_t)o * a.Ci + i) * M + k] = 0.5f * acc;
    }
}

}  // namespace hno

using namespace hno;

static int pm_check(const PmArgs &a) {
    HNO_REQUIRE(a.B > 0 && a.Ci > 0 && a.Co > 0 && a.M > 0, "hno_permode: bad size");
    if (a.B > 8) return fail(HNO_ELIMIT, "hno_permode: batch %d (max 8 per call)", a.B);
    if (a.Ci * a.Co > 65535) return fail(HNO_ELIMIT, "hno_permode: Ci*Co exceeds 65535");
    if (!a.fourier) HNO_REQUIRE(a.d0 * a.d1 * a.d2 == a.M, "hno_permode: mode grid does not match M");
    return HNO_OK;
}

extern "C" int hno_permode_fwd(const float *x, const float *xr, const float *w, const float *wi, float *y, int B, int Ci,
                               int Co, int M, int d0, int d1, int d2, int fourier, void *stream) {
    HNO_REQUIRE(x && w && y && (fourier ? wi != nullptr : xr != nullptr), "hno_permode_fwd: null pointer");
    PmArgs a = {};
    a.x = x; a.xr = xr; a.w = w; a.wi = wi; a.y = y;
    a.B = B; a.Ci = Ci; a.Co = Co; a.M = M; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.fourier = fourier;
    int rc = pm_check(a);
    if (rc) return rc;
    ProfScope _ps(KID_PERMODE_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(permode_fwd_kernel<8>, dim3(ceil_div(M, 256), Co), dim3(256), 0, (hipStream_t)stream, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_permode_bwd(const float *g, const float *x, const float *xr, const float *w, const float *wi, float *gx,
                               float *gxr, float *dw, float *dwi, int B, int Ci, int Co, int M, int d0, int d1, int d2,
                               int fourier, void *stream) {
    HNO_REQUIRE(g && x && w && gx && dw && (fourier ? (wi && dwi) : (xr && gxr)), "hno_permode_bwd: null pointer");
    PmArgs a = {};
    a.g = g; a.x = x; a.xr = xr; a.w = w; a.wi = wi; a.gx = gx; a.gxr = gxr; a.dw = dw; a.dwi = dwi;
    a.B = B; a.Ci = Ci; a.Co = Co; a.M = M; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.fourier = fourier;
    int rc = pm_check(a);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    ProfScope _ps(KID_PERMODE_BWD, s);
    hipLaunchKernelGGL(permode_dgrad_kernel<8>, dim3(ceil_div(M, 256), Ci), dim3(256), 0, s, a);
    HNO_CHECK_LAUNCH();
    hipLaunchKernelGGL(permode_wgrad_kernel<8>, dim3(ceil_div(M, 256), Ci * Co), dim3(256), 0, s, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
