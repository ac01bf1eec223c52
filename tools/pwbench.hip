// Standalone tuning bench for the 48->24 pointwise conv forward (act(W [xa;xb] + b)), odd row length.
// Variants of the access shape / tiling; prints us and algorithmic GB/s.  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) f2u { float x, y; };
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float selu(float x) {
    const float t = x * (1.f + x * (0.5f + x * (1.f / 6 + x * (1.f / 24 + x * (1.f / 120 + x * (1.f / 720 + x * (1.f / 5040)))))));
    const float e = __expf(x) - 1.f;
    const float m = x > -0.25f ? t : e;
    return x > 0.f ? 1.0507009873554805f * x : 1.7580993408473766f * m;
}
struct Args { const float *xa, *xb, *W, *bias; float *y; int B; unsigned V; int mode; };

// VEC consecutive voxels per lane; lane half h reads channel 2ks+h; tile = 32*VEC voxels per wave.
// NW waves per block, persistent grid-stride over tiles.  PF: software prefetch of the next tile.
template <int NKI, int COUT, int VEC, int NW, bool PF>
__global__ __launch_bounds__(64 * NW) void pw_vec(Args a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int CIN = 2 * NKI, CA = CIN / 2;
    const unsigned V = a.V;
    float w[NKI];
#pragma unroll
    for (int ks = 0; ks < NKI; ++ks) w[ks] = c < COUT ? a.W[c * CIN + 2 * ks + h] : 0.f;
    float bias_r[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) bias_r[j] = a.bias[(j & 3) + 8 * (j >> 2) + 4 * h];
    constexpr unsigned TV = 32 * VEC;
    const unsigned tiles_per_b = (V + TV - 1) / TV, ntiles = tiles_per_b * a.B;
    const unsigned stride = gridDim.x * NW;
    float x[NKI][VEC], xn[NKI][VEC];
    auto fetch = [&](unsigned t, float (&dst)[NKI][VEC]) {
        const unsigned b = t / tiles_per_b;
        const unsigned v0 = (t - b * tiles_per_b) * TV + c * VEC;
        const float *xa_b = a.xa + (size_t)b * CA * V, *xb_b = a.xb + (size_t)b * CA * V;
        if (v0 + VEC <= V) {
            const unsigned off = (h ? V : 0u) + v0;
#pragma unroll
            for (int ks = 0; ks < NKI; ++ks) {
                const int i0 = 2 * ks;
                const float *base = i0 < CA ? xa_b + (size_t)i0 * V : xb_b + (size_t)(i0 - CA) * V;
                if (VEC == 4) { const f4u q = *(const f4u *)(base + off); dst[ks][0] = q.x; dst[ks][1 % VEC] = q.y; dst[ks][2 % VEC] = q.z; dst[ks][3 % VEC] = q.w; }
                else if (VEC == 2) { const f2u q = *(const f2u *)(base + off); dst[ks][0] = q.x; dst[ks][1 % VEC] = q.y; }
                else dst[ks][0] = base[off];
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < NKI; ++ks) {
                const int i0 = 2 * ks;
                const float *base = (i0 < CA ? xa_b + (size_t)i0 * V : xb_b + (size_t)(i0 - CA) * V) + (h ? V : 0u);
#pragma unroll
                for (int r = 0; r < VEC; ++r) dst[ks][r] = v0 + r < V ? base[v0 + r] : 0.f;
            }
        }
    };
    unsigned t = blockIdx.x * NW + wave;
    if (PF && t < ntiles) fetch(t, x);
    for (; t < ntiles; t += stride) {
        const unsigned b = t / tiles_per_b;
        const unsigned v0 = (t - b * tiles_per_b) * TV + c * VEC;
        const bool full = v0 + VEC <= V;
        if (PF) { if (t + stride < ntiles) fetch(t + stride, xn); }
        else fetch(t, x);
        f32x16 acc[VEC];
#pragma unroll
        for (int r = 0; r < VEC; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
        if (a.mode & 1) {
#pragma unroll
            for (int ks = 0; ks < NKI; ++ks)
#pragma unroll
                for (int r = 0; r < VEC; ++r) acc[r][ks & 15] += x[ks][r] * w[ks];
        } else {
#pragma unroll
            for (int ks = 0; ks < NKI; ++ks)
#pragma unroll
                for (int r = 0; r < VEC; ++r) acc[r] = mfma32(w[ks], x[ks][r], acc[r]);
        }
        float *y_b = a.y + (size_t)b * COUT * V;
        const unsigned ooff = (h ? 4u * V : 0u) + v0;
        if (a.mode & 2) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < VEC; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) s += acc[r][j];
            if (s == 12345.678f) y_b[0] = s;
        } else if (full) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {   // COUT = 24: accumulator rows (j&3)+8(j>>2)+4h, j < 12
                const int orow = (j & 3) + 8 * (j >> 2);
                float *dst = y_b + (size_t)orow * V + ooff;
                if (VEC == 4) { f4u q; q.x = selu(acc[0][j] + bias_r[j]); q.y = selu(acc[1 % VEC][j] + bias_r[j]); q.z = selu(acc[2 % VEC][j] + bias_r[j]); q.w = selu(acc[3 % VEC][j] + bias_r[j]); *(f4u *)dst = q; }
                else if (VEC == 2) { f2u q; q.x = selu(acc[0][j] + bias_r[j]); q.y = selu(acc[1 % VEC][j] + bias_r[j]); *(f2u *)dst = q; }
                else dst[0] = selu(acc[0][j] + bias_r[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int orow = (j & 3) + 8 * (j >> 2);
#pragma unroll
                for (int r = 0; r < VEC; ++r)
                    if (v0 + r < V) (y_b + (size_t)orow * V + ooff)[r] = selu(acc[r][j] + bias_r[j]);
            }
        }
        if (PF) {
#pragma unroll
            for (int ks = 0; ks < NKI; ++ks)
#pragma unroll
                for (int r = 0; r < VEC; ++r) x[ks][r] = xn[ks][r];
        }
    }
}

// Blocked layout [B][C/8][V][8]: the 8 channels of a group are 32 contiguous bytes per voxel.  One wave tile = 64
// voxels: per input group two dwordx4 loads per lane (2 KB contiguous per instruction pair), v_permlane32_swap turns
// (channel pair of voxel l) registers into the two MFMA B operands (voxels 0..31 / 32..63); outputs are stored as
// dwordx4 per lane = 4 channels of one voxel, 1 KB contiguous per store instruction.
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void swap32(float &a, float &b) {
    // a = [a.lo | b.lo], b = [a.hi | b.hi]   (lo = lanes 0..31)
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int GIN, int GOUT, int NW>   // channel groups of 8: inputs (xa then xb), outputs
__global__ __launch_bounds__(64 * NW) void pw_blocked(Args a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int CIN = 8 * GIN, COUT = 8 * GOUT, NKI = CIN / 2, GA = GIN / 2;
    const unsigned V = a.V;
    float w[NKI];
#pragma unroll
    for (int ks = 0; ks < NKI; ++ks) w[ks] = c < COUT ? a.W[c * CIN + 2 * ks + h] : 0.f;
    float bias_r[4 * GOUT];
#pragma unroll
    for (int j = 0; j < 4 * GOUT; ++j) bias_r[j] = a.bias[(j & 3) + 8 * (j >> 2) + 4 * h];
    const unsigned tiles_per_b = (V + 63) / 64, ntiles = tiles_per_b * a.B;
    for (unsigned t = blockIdx.x * NW + wave; t < ntiles; t += gridDim.x * NW) {
        const unsigned b = t / tiles_per_b;
        const unsigned v = (t - b * tiles_per_b) * 64 + lane;
        const bool vin = v < V;
        const unsigned vl = vin ? v : 0u;
        f32x16 acc0, acc1;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.f;
        f4 lo[GIN], hi[GIN];
#pragma unroll
        for (int g = 0; g < GIN; ++g) {
            const float *base = (g < GA ? a.xa + ((size_t)b * GA + g) * V * 8 : a.xb + ((size_t)b * GA + (g - GA)) * V * 8) + (size_t)vl * 8;
            lo[g] = *reinterpret_cast<const f4 *>(base);
            hi[g] = *reinterpret_cast<const f4 *>(base + 4);
        }
#pragma unroll
        for (int g = 0; g < GIN; ++g) {
            float r[8] = {lo[g].x, lo[g].y, lo[g].z, lo[g].w, hi[g].x, hi[g].y, hi[g].z, hi[g].w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float e = r[2 * p], o = r[2 * p + 1];     // channels 8g+2p, 8g+2p+1 of voxel `lane`
                if (!(a.mode & 1)) {
                    swap32(e, o);                          // e: voxels 0..31 (k = h), o: voxels 32..63
                    acc0 = mfma32(w[4 * g + p], e, acc0);
                    acc1 = mfma32(w[4 * g + p], o, acc1);
                } else {
                    acc0[p] += e * w[4 * g + p];
                    acc1[p] += o * w[4 * g + p];
                }
            }
        }
        if (a.mode & 2) {
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) sacc += acc0[j] + acc1[j];
            if (sacc == 12345.678f) a.y[0] = sacc;
        } else {
            // acc register 4g'+i of lane (h, c): channel 8g' + 4h + i of voxel c (acc0) / 32 + c (acc1)
            const unsigned vb = (t - b * tiles_per_b) * 64;
#pragma unroll
            for (int g = 0; g < GOUT; ++g) {
                f4 o0, o1;
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    o0[i2] = selu(acc0[4 * g + i2] + bias_r[4 * g + i2]);
                    o1[i2] = selu(acc1[4 * g + i2] + bias_r[4 * g + i2]);
                }
                float *yb = a.y + (((size_t)b * GOUT + g) * V) * 8 + 4 * h;
                if (vb + c < V) *reinterpret_cast<f4 *>(yb + (size_t)(vb + c) * 8) = o0;
                if (vb + 32 + c < V) *reinterpret_cast<f4 *>(yb + (size_t)(vb + 32 + c) * 8) = o1;
            }
        }
    }
}

// Interleaved variant: the load of k-step ks of the NEXT tile is issued right before the MFMA of k-step ks of the
// current tile, so the memory pipe sees a steady stream instead of a 24-load burst followed by 1536 idle cycles.
template <int NKI, int COUT, int NW>
__global__ __launch_bounds__(64 * NW) void pw_inter(Args a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int CIN = 2 * NKI, CA = CIN / 2;
    const unsigned V = a.V;
    float w[NKI];
#pragma unroll
    for (int ks = 0; ks < NKI; ++ks) w[ks] = c < COUT ? a.W[c * CIN + 2 * ks + h] : 0.f;
    float bias_r[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) bias_r[j] = a.bias[(j & 3) + 8 * (j >> 2) + 4 * h];
    const unsigned tiles_per_b = (V + 31) / 32, ntiles = tiles_per_b * a.B;
    const unsigned stride = gridDim.x * NW;
    float x[NKI], xn[NKI];
    auto addr = [&](unsigned t, int ks) -> const float * {
        const unsigned b = t / tiles_per_b;
        const unsigned v = (t - b * tiles_per_b) * 32 + c;
        const unsigned off = (h ? V : 0u) + (v < V ? v : 0u);
        const int i0 = 2 * ks;
        const float *base = i0 < CA ? a.xa + (size_t)b * CA * V + (size_t)i0 * V : a.xb + (size_t)b * CA * V + (size_t)(i0 - CA) * V;
        return base + off;
    };
    unsigned t = blockIdx.x * NW + wave;
    if (t < ntiles) {
#pragma unroll
        for (int ks = 0; ks < NKI; ++ks) x[ks] = *addr(t, ks);
    }
    for (; t < ntiles; t += stride) {
        const unsigned b = t / tiles_per_b;
        const unsigned v = (t - b * tiles_per_b) * 32 + c;
        const bool vin = v < V;
        const bool more = t + stride < ntiles;
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKI; ++ks) {
            if (more) xn[ks] = *addr(t + stride, ks);
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32(w[ks], x[ks], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        float *y_b = a.y + (size_t)b * COUT * V;
        const unsigned ooff = (h ? 4u * V : 0u) + v;
        if (vin) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int orow = (j & 3) + 8 * (j >> 2);
                (y_b + (size_t)orow * V)[ooff] = selu(acc[j] + bias_r[j]);
            }
        }
#pragma unroll
        for (int ks = 0; ks < NKI; ++ks) x[ks] = xn[ks];
    }
}

template <typename K>
float timeit(K kern, int grid, int threads, Args a, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / reps * 1e3f;
}

int main(int argc, char **argv) {
    const unsigned V = argc > 1 ? (unsigned)atoi(argv[1]) : 65 * 65 * 65; const int B = 2, C = 24;
    printf("V = %u\n", V);
    const size_t n = (size_t)B * C * V;
    std::vector<float> hxa(n), hxb(n), hW(24 * 48), hb(24);
    srand(1);
    for (auto &v : hxa) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto &v : hxb) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto &v : hW) v = (rand() % 2001 - 1000) / 5000.f;
    for (auto &v : hb) v = (rand() % 2001 - 1000) / 10000.f;
    float *xa, *xb, *W, *bias, *y;
    hipMalloc(&xa, n * 4 + 256); hipMalloc(&xb, n * 4 + 256); hipMalloc(&y, n * 4 + 256); hipMalloc(&W, 24 * 48 * 4); hipMalloc(&bias, 24 * 4);
    hipMemcpy(xa, hxa.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(xb, hxb.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), 24 * 48 * 4, hipMemcpyHostToDevice); hipMemcpy(bias, hb.data(), 24 * 4, hipMemcpyHostToDevice);
    Args a{xa, xb, W, bias, y, B, V, 0};
    const double mb = 4.0 * B * V * (48 + 24) / 1e6;
    // correctness of each variant on a sample of voxels
    auto check = [&](const char *name) {
        std::vector<float> hy(n);
        hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost);
        double maxe = 0;
        for (int s = 0; s < 4000; ++s) {
            const unsigned b = s & 1, v = (s < 200) ? V - 1 - s : (unsigned)((size_t)s * 7919 % V);
            for (int o = 0; o < 24; ++o) {
                double acc = hb[o];
                for (int i = 0; i < 48; ++i) acc += (double)hW[o * 48 + i] * (i < 24 ? hxa[((size_t)b * 24 + i) * V + v] : hxb[((size_t)b * 24 + i - 24) * V + v]);
                const double ref = acc > 0 ? 1.0507009873554805 * acc : 1.7580993408473766 * expm1(acc);
                maxe = fmax(maxe, fabs(ref - hy[((size_t)b * 24 + o) * V + v]));
            }
        }
        printf("  err %.1e\n", maxe);
    };
#define RUN(name, VEC, NW, PF, grid)                                                                  \
    {                                                                                                  \
        auto kern = pw_vec<24, 24, VEC, NW, PF>;                                                       \
        a.mode = 0; float t = timeit(kern, grid, 64 * NW, a);                                          \
        a.mode = 1; float t1 = timeit(kern, grid, 64 * NW, a);                                         \
        a.mode = 2; float t2 = timeit(kern, grid, 64 * NW, a);                                         \
        a.mode = 3; float t3 = timeit(kern, grid, 64 * NW, a);                                         \
        a.mode = 0; hipMemset(y, 0, n * 4); hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), 0, 0, a);  \
        printf("%-28s vec%d nw%2d pf%d grid %4d: full %6.1f us (%5.0f GB/s)  no-mfma %6.1f  no-store %6.1f  loads-only %6.1f", name, VEC, NW, (int)PF, grid, t, mb / t * 1e3, t1, t2, t3); \
        check(name);                                                                                   \
    }
    RUN("", 1, 8, false, 256) RUN("", 1, 4, false, 512)
#define RUNI(NW, grid)                                                                                 \
    {                                                                                                  \
        auto kern = pw_inter<24, 24, NW>;                                                              \
        a.mode = 0; float t = timeit(kern, grid, 64 * NW, a);                                          \
        hipMemset(y, 0, n * 4); hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), 0, 0, a);          \
        printf("interleaved nw%2d grid %4d: full %6.1f us (%5.0f GB/s)", NW, grid, t, mb / t * 1e3);    \
        check("");                                                                                     \
    }
    RUNI(8, 256) RUNI(4, 256) RUNI(4, 512) RUNI(8, 512) RUNI(12, 256)
    return 0;
}
