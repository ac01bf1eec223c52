// Multi-tensor Adamax: one launch updates every parameter of the model (the reference drives torch.optim.Adamax,
// experiments/run.py:89-91, config_files/config_hnoseg_xs.ini:53-55; the per-element arithmetic below is the
// published algorithm as torch implements it, torch/optim/adamax.py `_single_tensor_adamax`).
//   g    = grad * grad_scale + weight_decay * p
//   m    = m + (1 - beta1) * (g - m)
//   u    = max(beta2 * u, |g| + eps)
//   p    = p - (lr / (1 - beta1^t)) * m / u
// HBM-bound streaming: 16 B read + 12 B written per element; HNOSeg-XS has 28 248 elements, so the point of the
// kernel is ONE launch instead of ~10 multi-tensor launches over ~60 tensors.
#include "hno_common.h"

namespace hno {

struct AdamaxChunk {   // one row of the table: a run of <= chunk elements of one tensor
    float *p;
    const float *g;
    float *m;
    float *u;
    long long n;
};

__global__ __launch_bounds__(256) void adamax_multi_kernel(const AdamaxChunk *__restrict__ table, float clr, float beta1,
                                                           float beta2, float eps, float wd, float gscale) {
    const AdamaxChunk c = table[blockIdx.x];
    for (int i = threadIdx.x; i < c.n; i += 256) {
        const float p = c.p[i];
        const float g = fmaf(wd, p, c.g[i] * gscale);
        float m = c.m[i];
        m = fmaf(1.f - beta1, g - m, m);
        const float u = fmaxf(beta2 * c.u[i], fabsf(g) + eps);
        c.m[i] = m;
        c.u[i] = u;
        c.p[i] = p - clr * (m / u);
    }
}

}  // namespace hno

using namespace hno;

extern "C" int hno_adamax_chunk_rows(void) { return (int)(sizeof(AdamaxChunk) / sizeof(long long)); }

extern "C" int hno_adamax_multi(const void *table, int n_chunks, float lr, float beta1, float beta2, float eps,
                                float weight_decay, long long step, float grad_scale, void *stream) {
    HNO_REQUIRE(table && n_chunks > 0 && step >= 1, "hno_adamax_multi: bad argument");
    HNO_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "hno_adamax_multi: bad hyper-parameter");
    const double bias_correction = 1.0 - pow((double)beta1, (double)step);
    const float clr = (float)((double)lr / bias_correction);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adamax_multi_kernel, dim3(n_chunks), dim3(256), 0, s, (const AdamaxChunk *)table, clr, beta1, beta2, eps,
                       weight_decay, grad_scale);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
