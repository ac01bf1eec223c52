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

// ---- device-stepped form (round 4): the step counter, the learning-rate schedule and the bias correction live in a small device
// state, so the launch has NO per-step host argument and the update can be captured into the training step's HIP graph (a rank's
// whole step is then one graph replay).  state (doubles; integers are exact):
//   [0] step (optimizer steps taken)   [1] lr (of the NEXT update)   [2] base_lr   [3] eta_min   [4] T_cur   [5] T_i   [6] T_mult
//   [7] schedule: 0 = constant lr, 1 = CosineAnnealingWarmRestarts stepped once per optimizer step (experiments/run.py:92-103,
//       train_test.py:173-174)          [8] clr = lr / (1 - beta1^(step + 1)) of the NEXT update          [9] ticket counter (bits)
//   [10] scheduler ticks (differs from [0] only by the updates GradScaler skipped)
//
// The state is advanced by the LAST workgroup of the update kernel itself (a ticket counter behind the doubles; round 4b: a separate
// one-thread kernel in front of the update cost a 5 us launch per step): every workgroup reads clr = st[8] at its start, takes a ticket
// when its stores are done, and the holder of the last ticket does   step += 1;   the scheduler's step() for the next update, as
// torch.optim.lr_scheduler.CosineAnnealingWarmRestarts.step() does it: T_cur += 1; if T_cur >= T_i: T_cur -= T_i, T_i *= T_mult;
// lr = eta_min + (base_lr - eta_min) (1 + cos(pi T_cur / T_i)) / 2, all in double;   st[8] = (float) lr / (1 - beta1^(step + 1)) for the
// NEXT update (the host writes the first one: optim.Adamax.device_stepped);   ticket counter = 0.
// skipped (round 6, GradScaler's found_inf): the update did not happen -- the optimizer's step count stays, the schedule still ticks (the
// reference steps its scheduler after every batch, experiments/train_test.py:173-174, whether or not GradScaler.step() ran the optimizer)
__device__ __forceinline__ void adamax_advance(double *st, double beta1, bool skipped = false) {
    const double step = st[0] + (skipped ? 0.0 : 1.0);
    st[0] = step;
    st[10] += 1.0;            // scheduler ticks (= the scheduler's last_epoch)
    if (st[7] == 1.0) {
        double T_cur = st[4] + 1.0, T_i = st[5];
        if (T_cur >= T_i) {
            T_cur -= T_i;
            T_i *= st[6];
        }
        st[4] = T_cur;
        st[5] = T_i;
        st[1] = st[3] + (st[2] - st[3]) * (1.0 + cos(M_PI * T_cur / T_i)) / 2.0;
    }
    st[8] = (double)(float)st[1] / (1.0 - pow(beta1, step + 1.0));      // (the eager entry point takes lr as a float: same rounding)
}

// amp_scale / found_inf (round 6): the two device scalars torch.amp.GradScaler hands an optimizer that declares
// `_step_supports_amp_scaling` -- the gradients are still multiplied by *amp_scale and are divided here (by the float reciprocal taken
// in double, the inv_scale GradScaler.unscale_ multiplies with; they are written back unscaled, as unscale_ leaves them), and a non-zero
// *found_inf skips the whole update: no host reads either value, so GradScaler.step() / update() can sit inside the captured step.
__global__ __launch_bounds__(256) void adamax_multi_dev_kernel(const AdamaxChunk *__restrict__ table, double *st, float beta1,
                                                               float beta2, float eps, float wd, float gscale,
                                                               const float *__restrict__ amp_scale, const float *__restrict__ found_inf) {
    const AdamaxChunk c = table[blockIdx.x];
    const float clr = (float)st[8];
    const bool skip = found_inf && *found_inf != 0.f;
    if (!skip) {
        const float inv = amp_scale ? (float)(1.0 / (double)*amp_scale) : 1.f;
        for (int i = threadIdx.x; i < c.n; i += 256) {
            const float p = c.p[i];
            float gr = c.g[i];
            if (amp_scale) {
                gr *= inv;
                const_cast<float *>(c.g)[i] = gr;
            }
            const float g = fmaf(wd, p, gr * gscale);
            float m = c.m[i];
            m = fmaf(1.f - beta1, g - m, m);
            const float u = fmaxf(beta2 * c.u[i], fabsf(g) + eps);
            c.m[i] = m;
            c.u[i] = u;
            c.p[i] = p - clr * (m / u);
        }
    }
    __syncthreads();                                       // (every thread of this workgroup has read clr)
    if (threadIdx.x == 0) {
        unsigned *ticket = reinterpret_cast<unsigned *>(st + 9);
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {      // all workgroups are past their read of st[8]
            adamax_advance(st, (double)beta1, skip);
            *ticket = 0u;
            __threadfence();
        }
    }
}


// ---- the join of two gradient sets (round 5): dst_i = scale_i * (a_i + b_i) for up to 64 tensors per launch -----------------------
// A captured step that runs the two halves of its batch as two passes (experiments.train_test.SampleSplit) ends with the halves'
// gradients in two tensor sets and two loss scalars.  torch._foreach_add_ + torch.lerp did the join in round 4: two ATen launches in a
// step that otherwise consists of libhno kernels only.  The tensors' addresses are known only while the step is being captured (nothing
// may be allocated there), so the entries travel BY VALUE in the kernel arguments (2 KB for 64 entries) and become part of the graph node.
#define HNO_MAX_PAIRS 64
struct PairEntry {
    float *dst;
    const float *a, *b;
    int n, first_block;
    float scale;
    int pad_;
};
struct PairBatch {
    PairEntry e[HNO_MAX_PAIRS];
    int count;
};
__global__ __launch_bounds__(256) void sum_pairs_kernel(PairBatch pb) {
    // one workgroup per tensor (the tensors of a model are small: the largest of HNOSeg-XS has 1 152 elements).  The first version mapped
    // 1 024-element blocks to entries by a linear search through the kernel arguments: 13.8 us for 28 k elements, all of it the search.
    const PairEntry &e = pb.e[blockIdx.x];
    for (int i = threadIdx.x; i < e.n; i += 256) e.dst[i] = e.scale * (e.a[i] + e.b[i]);
}

}  // namespace hno

using namespace hno;

extern "C" int hno_adamax_state_doubles(void) { return 11; }   // 9 doubles + the ticket counter + the scheduler's tick count

// one Adamax step driven by the device state (see above); capturable: no host value changes from step to step
extern "C" int hno_adamax_multi_dev(const void *table, int n_chunks, void *state, float beta1, float beta2, float eps, float weight_decay,
                                    float grad_scale, void *stream) {
    HNO_REQUIRE(table && n_chunks > 0 && state, "hno_adamax_multi_dev: bad argument");
    HNO_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "hno_adamax_multi_dev: bad hyper-parameter");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adamax_multi_dev_kernel, dim3(n_chunks), dim3(256), 0, s, (const AdamaxChunk *)table, (double *)state, beta1, beta2,
                       eps, weight_decay, grad_scale, (const float *)nullptr, (const float *)nullptr);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// the same under torch.amp.GradScaler (see adamax_multi_dev_kernel): amp_scale / found_inf are DEVICE scalars (either may be NULL)
extern "C" int hno_adamax_multi_dev_amp(const void *table, int n_chunks, void *state, float beta1, float beta2, float eps, float weight_decay,
                                        float grad_scale, const float *amp_scale, const float *found_inf, void *stream) {
    HNO_REQUIRE(table && n_chunks > 0 && state, "hno_adamax_multi_dev_amp: bad argument");
    HNO_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "hno_adamax_multi_dev_amp: bad hyper-parameter");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adamax_multi_dev_kernel, dim3(n_chunks), dim3(256), 0, s, (const AdamaxChunk *)table, (double *)state, beta1, beta2,
                       eps, weight_decay, grad_scale, amp_scale, found_inf);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

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

// dst[i] = scale[i] * (a[i] + b[i]) elementwise for `count` fp32 tensors of n[i] elements (HOST arrays of device pointers / sizes);
// dst may alias a or b.  One launch per 64 tensors; capturable (no device table: the entries are kernel arguments).
extern "C" int hno_sum_pairs(void *const *dst, const void *const *a, const void *const *b, const long long *n, const float *scale,
                             int count, void *stream) {
    HNO_REQUIRE(dst && a && b && n && scale && count > 0, "hno_sum_pairs: bad argument");
    hipStream_t s = (hipStream_t)stream;
    constexpr long long CHUNK = 1 << 16;      // elements per workgroup: large tensors become several entries
    PairBatch pb;
    pb.count = 0;
    auto flush = [&]() {
        if (pb.count) hipLaunchKernelGGL(sum_pairs_kernel, dim3(pb.count), dim3(256), 0, s, pb);
        pb.count = 0;
    };
    for (int i = 0; i < count; ++i) {
        HNO_REQUIRE(dst[i] && a[i] && b[i] && n[i] >= 0, "hno_sum_pairs: bad entry");
        for (long long off = 0; off < n[i]; off += CHUNK) {
            const long long m = n[i] - off < CHUNK ? n[i] - off : CHUNK;
            pb.e[pb.count++] = PairEntry{(float *)dst[i] + off, (const float *)a[i] + off, (const float *)b[i] + off, (int)m, 0, scale[i], 0};
            if (pb.count == HNO_MAX_PAIRS) flush();
        }
    }
    flush();
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
