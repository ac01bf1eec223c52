// GPU-side input pipeline: what the reference runs per sample in CPU DataLoader
// workers before a batch reaches the model.
//   * hno_zscore_modalities -- experiments/utils.py:25-71 (normalize_modalities / normalize_data): per modality,
//     optional clip, mean / population std over the voxels != mask_val, (x - mean) / std, masked voxels -> 0.
//   * hno_affine_nearest    -- experiments/data_io/dataset.py:205-237 (apply_transform) + :240-244 (flip_axis): nearest
//     neighbour resampling through an affine map, constant fill outside, optional axis flips.  The reference
//     delegates this to SimpleITK (sitk.AffineTransform + ResampleImageFilter, sitkNearestNeighbor), which is not
//     in this image: the kernel restates ITK's published semantics (continuous index = M p + t in double precision,
//     inside test [-0.5, size - 0.5), round-half-up) -- PARITY UNPINNED for this kernel (no reference run possible).
// Both are pure streaming / gather kernels (HBM bound): z-score reads x twice and writes once (12 B per voxel),
// the resampler reads <= 4 B and writes 4 B per voxel.
#include "hno_common.h"

namespace hno {

constexpr int ZS_BLOCKS = 64;     // partial-sum blocks per modality
constexpr int ZS_THREADS = 256;

__device__ __forceinline__ float zs_clip(float v, int has_clip, float lo, float hi) {
    return has_clip ? fminf(fmaxf(v, lo), hi) : v;
}

// partial (count, sum, sum of squares) per (modality, block) in double; squares are taken about a pilot value
// (the first voxel of the modality) so that one pass is as accurate as numpy's two-pass std for MR intensities
__global__ __launch_bounds__(ZS_THREADS) void zscore_stats_kernel(const float *__restrict__ x, double *__restrict__ part,
                                                                   long long V, int has_mask, float mask_val, int has_clip,
                                                                   float lo, float hi) {
    const int c = blockIdx.y;
    const float *xc = x + (size_t)c * V;
    const double pilot = (double)zs_clip(xc[0], has_clip, lo, hi);
    double n = 0.0, s = 0.0, q = 0.0;
    for (long long i = (long long)blockIdx.x * ZS_THREADS + threadIdx.x; i < V; i += (long long)ZS_BLOCKS * ZS_THREADS) {
        const float v = zs_clip(xc[i], has_clip, lo, hi);
        if (has_mask && v == mask_val) continue;
        const double d = (double)v - pilot;
        n += 1.0;
        s += d;
        q += d * d;
    }
    __shared__ double red[3][ZS_THREADS / 64];
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off);
        s += __shfl_down(s, off);
        q += __shfl_down(q, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = n;
        red[1][wave] = s;
        red[2][wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = part + ((size_t)c * ZS_BLOCKS + blockIdx.x) * 3;
        o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        o[2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

__global__ __launch_bounds__(ZS_THREADS) void zscore_apply_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                                   const double *__restrict__ part, float *__restrict__ stats,
                                                                   long long V, int has_mask, float mask_val, int has_clip,
                                                                   float lo, float hi) {
    const int c = blockIdx.y;
    const float *xc = x + (size_t)c * V;
    float *oc = out + (size_t)c * V;
    // every workgroup re-reduces the 64 partials of its modality in the same fixed order (deterministic)
    double n = 0.0, s = 0.0, q = 0.0;
    for (int b = 0; b < ZS_BLOCKS; ++b) {
        const double *p = part + ((size_t)c * ZS_BLOCKS + b) * 3;
        n += p[0];
        s += p[1];
        q += p[2];
    }
    const double pilot = (double)zs_clip(xc[0], has_clip, lo, hi);
    const double md = s / n;                       // mean - pilot
    double var = q / n - md * md;                  // population variance (numpy std, ddof = 0)
    if (var < 0.0) var = 0.0;
    const float mean = (float)(pilot + md), stdv = (float)sqrt(var);   // std = 0 -> inf / nan like numpy
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) {
        stats[2 * c] = mean;
        stats[2 * c + 1] = stdv;
    }
    for (long long i = (long long)blockIdx.x * ZS_THREADS + threadIdx.x; i < V; i += (long long)gridDim.x * ZS_THREADS) {
        const float v = zs_clip(xc[i], has_clip, lo, hi);
        const bool masked = has_mask && v == mask_val;
        __builtin_nontemporal_store(masked ? 0.f : (v - mean) / stdv, oc + i);
    }
}

struct AffineArgs {
    double m[12];   // rows of [M | t]: input continuous index (x, y, z) = M * output index (x, y, z) + t
    int C, D, H, W;
    int flip;       // bit 0: depth, bit 1: height, bit 2: width -- applied to the RESAMPLED image
    float cval;
};

__global__ __launch_bounds__(256) void affine_nearest_kernel(const float *__restrict__ x, float *__restrict__ out, AffineArgs a) {
    const long long V = (long long)a.D * a.H * a.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
        const int w = (int)(i % a.W);
        const long long r = i / a.W;
        const int h = (int)(r % a.H), d = (int)(r / a.H);
        // output voxel (d, h, w) shows resampled voxel (d', h', w') after the flips
        const double px = (a.flip & 4) ? a.W - 1 - w : w, py = (a.flip & 2) ? a.H - 1 - h : h, pz = (a.flip & 1) ? a.D - 1 - d : d;
        const double cx = a.m[0] * px + a.m[1] * py + a.m[2] * pz + a.m[3];
        const double cy = a.m[4] * px + a.m[5] * py + a.m[6] * pz + a.m[7];
        const double cz = a.m[8] * px + a.m[9] * py + a.m[10] * pz + a.m[11];
        const bool inside = cx >= -0.5 && cx < a.W - 0.5 && cy >= -0.5 && cy < a.H - 0.5 && cz >= -0.5 && cz < a.D - 0.5;
        const long long src = inside ? ((long long)floor(cz + 0.5) * a.H + (long long)floor(cy + 0.5)) * a.W + (long long)floor(cx + 0.5) : 0;
        for (int c = 0; c < a.C; ++c) {
            const float v = inside ? x[(size_t)c * V + src] : a.cval;
            __builtin_nontemporal_store(v, out + (size_t)c * V + i);
        }
    }
}

}  // namespace hno

using namespace hno;

extern "C" size_t hno_zscore_workspace_bytes(int C) { return C > 0 ? (size_t)C * ZS_BLOCKS * 3 * sizeof(double) : 0; }

extern "C" int hno_zscore_modalities(const float *x, float *out, float *mean_std, void *workspace, int C, long long V,
                                     int has_mask, float mask_val, int has_clip, float clip_lo, float clip_hi, void *stream) {
    HNO_REQUIRE(x && out && workspace && C > 0 && V > 0, "hno_zscore_modalities: bad argument");
    if (C > 65535) return fail(HNO_ELIMIT, "hno_zscore_modalities: %d modalities (max 65535)", C);
    hipStream_t s = (hipStream_t)stream;
    double *part = (double *)workspace;
    hipLaunchKernelGGL(zscore_stats_kernel, dim3(ZS_BLOCKS, C), dim3(ZS_THREADS), 0, s, x, part, V, has_mask, mask_val, has_clip,
                       clip_lo, clip_hi);
    long long nb = (V + ZS_THREADS * 8 - 1) / (ZS_THREADS * 8);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(zscore_apply_kernel, dim3((unsigned)nb, C), dim3(ZS_THREADS), 0, s, x, out, (const double *)part, mean_std, V,
                       has_mask, mask_val, has_clip, clip_lo, clip_hi);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_affine_nearest(const float *x, float *out, const double *matrix12, float cval, int flip_mask, int C, int D,
                                  int H, int W, void *stream) {
    HNO_REQUIRE(x && out && matrix12 && C > 0 && D > 0 && H > 0 && W > 0, "hno_affine_nearest: bad argument");
    HNO_REQUIRE(x != out, "hno_affine_nearest: in-place resampling is not possible");
    AffineArgs a;
    for (int i = 0; i < 12; ++i) a.m[i] = matrix12[i];
    a.C = C, a.D = D, a.H = H, a.W = W, a.flip = flip_mask, a.cval = cval;
    const long long V = (long long)D * H * W;
    long long nb = (V + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(affine_nearest_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, out, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
