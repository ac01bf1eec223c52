// Ceiling experiment for the pointwise kernels: read CIN planar streams, write COUT planar streams (out[o] = in[2o] + in[2o+1]),
// with different access shapes per wave instruction.  Build: hipcc --offload-arch=gfx950 -O3 -o planar_copy planar_copy.hip
//   mode 0: 2 rows x 32 voxels x 4 B per instruction (what the MFMA operand layout of the shipped kernel gives)
//   mode 1: 1 row x 64 voxels x 4 B
//   mode 2: 1 row x 256 voxels x 16 B per lane (aligned; V % 4 == 0 needed)
//   mode 3: as mode 2 but every channel row starts at the 16-byte boundary below its (unaligned) address: V odd allowed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int CIN = 48, COUT = 24;

template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void planar_kernel(const float *__restrict__ x, float *__restrict__ y, int B, unsigned V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (MODE == 0) {
        const int h = lane >> 5, c = lane & 31;
        const unsigned tpb = (V + 31) / 32, nt = tpb * B;
        for (unsigned t = blockIdx.x * NW + wave; t < nt; t += gridDim.x * NW) {
            const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
            const bool in = v < V;
            const float *xb = x + (size_t)b * CIN * V + (in ? v : 0u) + (h ? V : 0u);
            float xv[CIN / 2];
#pragma unroll
            for (int k = 0; k < CIN / 2; ++k) xv[k] = xb[(size_t)2 * k * V];
            float *yb = y + (size_t)b * COUT * V + v + (h ? V : 0u);
#pragma unroll
            for (int o = 0; o < COUT / 2; ++o) {
                const float s = xv[2 * o] + xv[2 * o + 1];   // rows 4o+h.. (not the real pairing; same traffic)
                if (in) yb[(size_t)2 * o * V] = s;
            }
        }
    } else if constexpr (MODE == 12) {
        // mode 0's access shape, but every wave walks a CONTIGUOUS range of tiles: the 32-byte sectors at the two ends of a
        // misaligned 128-byte run are shared with the neighbouring tiles, which the same wave requests next (round 3, pmc_planar.sh)
        const int h = lane >> 5, c = lane & 31;
        const unsigned tpb = (V + 31) / 32, nt = tpb * B;
        const unsigned gw = blockIdx.x * NW + wave, nwv = gridDim.x * NW;
        const unsigned per = nt / nwv, rem = nt - per * nwv;
        const unsigned t0 = gw * per + (gw < rem ? gw : rem), t1 = t0 + per + (gw < rem ? 1u : 0u);
        for (unsigned t = t0; t < t1; ++t) {
            const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
            const bool in = v < V;
            const float *xb = x + (size_t)b * CIN * V + (in ? v : 0u) + (h ? V : 0u);
            float xv[CIN / 2];
#pragma unroll
            for (int k = 0; k < CIN / 2; ++k) xv[k] = xb[(size_t)2 * k * V];
            float *yb = y + (size_t)b * COUT * V + v + (h ? V : 0u);
#pragma unroll
            for (int o = 0; o < COUT / 2; ++o) {
                const float s = xv[2 * o] + xv[2 * o + 1];
                if (in) yb[(size_t)2 * o * V] = s;
            }
        }
    } else if constexpr (MODE == 1) {
        const unsigned tpb = (V + 63) / 64, nt = tpb * B;
        for (unsigned t = blockIdx.x * NW + wave; t < nt; t += gridDim.x * NW) {
            const unsigned b = t / tpb, v = (t - b * tpb) * 64 + lane;
            const bool in = v < V;
            const float *xb = x + (size_t)b * CIN * V + (in ? v : 0u);
            float xv[CIN];
#pragma unroll
            for (int k = 0; k < CIN; ++k) xv[k] = xb[(size_t)k * V];
            float *yb = y + (size_t)b * COUT * V + v;
#pragma unroll
            for (int o = 0; o < COUT; ++o)
                if (in) yb[(size_t)o * V] = xv[2 * o] + xv[2 * o + 1];
        }
    } else if constexpr (MODE == 2) {
        const unsigned tpb = (V + 255) / 256, nt = tpb * B;
        for (unsigned t = blockIdx.x * NW + wave; t < nt; t += gridDim.x * NW) {
            const unsigned b = t / tpb, v = (t - b * tpb) * 256 + 4 * lane;
            const bool in = v < V;
            const float *xb = x + (size_t)b * CIN * V + (in ? v : 0u);
            float *yb = y + (size_t)b * COUT * V + v;
#pragma unroll
            for (int g = 0; g < COUT; g += 8) {      // 16 input rows at a time: 64 VGPRs of loads in flight
                float4 xv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) xv[k] = *(const float4 *)(xb + (size_t)(2 * g + k) * V);
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    float4 s;
                    s.x = xv[2 * o].x + xv[2 * o + 1].x; s.y = xv[2 * o].y + xv[2 * o + 1].y;
                    s.z = xv[2 * o].z + xv[2 * o + 1].z; s.w = xv[2 * o].w + xv[2 * o + 1].w;
                    if (in) *(float4 *)(yb + (size_t)(g + o) * V) = s;
                }
            }
        }
    } else {
        // every row is read / written in aligned 16-byte chunks from the boundary below its own start: the window of row k covers
        // elements [e0 - (e0 & 3), ...) where e0 is the global element index of the row's first voxel of the tile
        const unsigned tpb = (V + 255) / 256, nt = tpb * B;
        for (unsigned t = blockIdx.x * NW + wave; t < nt; t += gridDim.x * NW) {
            const unsigned b = t / tpb, v0 = (t - b * tpb) * 256;
#pragma unroll
            for (int g = 0; g < COUT; g += 8) {
                float4 xv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const size_t e0 = ((size_t)b * CIN + 2 * g + k) * V + v0;
                    size_t e = (e0 & ~(size_t)3) + 4 * lane;
                    const size_t lim = (size_t)B * CIN * V - 4;
                    xv[k] = *(const float4 *)(x + (e < lim ? e : lim));
                }
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    float4 s;
                    s.x = xv[2 * o].x + xv[2 * o + 1].x; s.y = xv[2 * o].y + xv[2 * o + 1].y;
                    s.z = xv[2 * o].z + xv[2 * o + 1].z; s.w = xv[2 * o].w + xv[2 * o + 1].w;
                    const size_t e0 = ((size_t)b * COUT + g + o) * V + v0;
                    const size_t e = (e0 & ~(size_t)3) + 4 * lane;
                    if (e + 4 <= (size_t)B * COUT * V) *(float4 *)(y + e) = s;     // (values are shifted garbage: traffic experiment only)
                }
            }
        }
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode 4: the shipped kernel's skeleton -- loads (2 x 128 B per instruction), 24 chained fp32 MFMAs, 12 stores.
// mode 5: the same with the loads as LDS-DMA into a wave-private two-slot ring, one tile ahead; B operands by ds_read_b32.
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void planar_mfma_kernel(const float *__restrict__ x, const float *__restrict__ W, float *__restrict__ y, int B, unsigned V, int abl = 0) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int NKI = CIN / 2;
    float w[NKI];
#pragma unroll
    for (int k = 0; k < NKI; ++k) w[k] = c < COUT ? W[c * CIN + 2 * k + h] : 0.f;
    const unsigned tpb = (V + 31) / 32, nt = tpb * B;
    const unsigned stride = gridDim.x * NW;
    if constexpr (MODE == 4) {
        for (unsigned t = blockIdx.x * NW + wave; t < nt; t += stride) {
            const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
            const bool in = v < V;
            const float *xb = x + (size_t)b * CIN * V + (in ? v : 0u) + (h ? V : 0u);
            float xv[NKI];
#pragma unroll
            for (int k = 0; k < NKI; ++k) xv[k] = xb[(size_t)2 * k * V];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int k = 0; k < NKI; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k], xv[k], acc, 0, 0, 0);
            float *yb = y + (size_t)b * COUT * V + v + (h ? 4 * V : 0u);
#pragma unroll
            for (int r = 0; r < 12; ++r)
                if (in) yb[(size_t)((r & 3) + 8 * (r >> 2)) * V] = acc[r];
        }
    } else {
        // wave-private ring: 2 slots x NKI x 64 floats
        constexpr int D = MODE == 5 ? 2 : 3;
        float *ring = lds + wave * (D * NKI * 64);
        const unsigned ring_b = (unsigned)(size_t)ring;   // LDS byte address (low 32 bits of the generic pointer)
        auto issue = [&](unsigned t, int slot) {
            const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
            const float *xb = x + (size_t)b * CIN * V + (v < V ? v : 0u) + (h ? V : 0u);
#pragma unroll
            for (int k = 0; k < NKI; ++k) {
                const float *src = xb + (size_t)2 * k * V;
                const unsigned dst = __builtin_amdgcn_readfirstlane(ring_b + (slot * NKI + k) * 256);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        };
        unsigned t = blockIdx.x * NW + wave;
        int slot = 0;
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (t + d * stride < nt) issue(t + d * stride, d);
        for (; t < nt; t += stride, slot = slot + 1 == D ? 0 : slot + 1) {
            const unsigned tn = t + (D - 1) * stride;
            if (tn < nt) {
                issue(tn, slot == 0 ? D - 1 : slot - 1);
                if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");      // 72 issued after this tile's DMA; 63 is the counter's limit
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
            const bool in = v < V;
            const float *sl = ring + slot * NKI * 64 + lane;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (abl & 1) {
#pragma unroll
                for (int k = 0; k < NKI; ++k) acc[k & 15] += sl[k * 64];
            } else {
#pragma unroll
                for (int k = 0; k < NKI; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k], sl[k * 64], acc, 0, 0, 0);
            }
            float *yb = y + (size_t)b * COUT * V + v + (h ? 4 * V : 0u);
            if (abl & 2) {
                float sacc = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[r];
                if (sacc == 12345.678f) yb[0] = sacc;
            } else {
#pragma unroll
                for (int r = 0; r < 12; ++r)
                    if (in) yb[(size_t)((r & 3) + 8 * (r >> 2)) * V] = acc[r];
            }
        }
    }
}

template <int MODE, int NW>
static float run_mfma(const float *x, const float *W, float *y, int B, unsigned V, int grid, int reps, int abl = 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = MODE >= 5 ? (size_t)NW * (MODE == 5 ? 2 : 3) * (CIN / 2) * 64 * 4 : 0;
    CK(hipFuncSetAttribute((const void *)planar_mfma_kernel<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((planar_mfma_kernel<MODE, NW>), dim3(grid), dim3(64 * NW), lds, 0, x, W, y, B, V, abl);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((planar_mfma_kernel<MODE, NW>), dim3(grid), dim3(64 * NW), lds, 0, x, W, y, B, V, abl);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

// mode 7: 16-byte LDS-DMA, 7 rows x 9 chunks per instruction; every row's window starts at the 16-byte boundary below its own
// first element (rows are only 4-byte aligned when V is odd), the consumer adds the row's misalignment to its LDS address.
template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void planar_dma16_kernel(const float *__restrict__ x, const float *__restrict__ W, float *__restrict__ y, int B, unsigned V) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int NKI = CIN / 2, NJ = (CIN + 6) / 7, SLOT = NJ * 256;   // floats per ring slot
    float w[NKI];
#pragma unroll
    for (int k = 0; k < NKI; ++k) w[k] = c < COUT ? W[c * CIN + 2 * k + h] : 0.f;
    const unsigned tpb = (V + 31) / 32, nt = tpb * B;
    const unsigned stride = gridDim.x * NW;
    float *ring = lds + wave * (D * SLOT);
    const unsigned ring_b = (unsigned)(size_t)ring;
    const unsigned xmis = (unsigned)((size_t)x >> 2) & 3u;
    // producer: per-lane byte offset of DMA j relative to (x + (b CIN V + v0) * 4)
    unsigned poff[NJ];
    {
        const int rj = lane / 9, chunk = lane - 9 * rj;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int row = 7 * j + rj;
            int ch = chunk;
            if (rj >= 7 || row >= CIN) { row = 7 * j; ch = 0; }     // idle lanes repeat a valid address; their bytes land in padding
            const unsigned m = (xmis + (unsigned)row * V) & 3u;
            poff[j] = ((unsigned)row * V - m + 4u * ch) * 4u;
        }
    }
    // consumer: LDS byte offset of k-step k's operand inside a slot
    unsigned coff[NKI];
#pragma unroll
    for (int k = 0; k < NKI; ++k) {
        const int row = 2 * k + h, j = row / 7, rj = row - 7 * j;
        const unsigned m = (xmis + (unsigned)row * V) & 3u;
        coff[k] = (j * 256 + rj * 36 + m + c) * 4u;
    }
    auto issue = [&](unsigned t, int slot) {
        const unsigned b = t / tpb, v0 = (t - b * tpb) * 32;
        const float *base = x + ((size_t)b * CIN * V + v0);           // wave-uniform
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(ring_b + (slot * SLOT + j * 256) * 4);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(poff[j]), "s"(base), "s"(dst) : "memory");
        }
    };
    unsigned t = blockIdx.x * NW + wave;
    int slot = 0;
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (t + d * stride < nt) issue(t + d * stride, d);
    for (; t < nt; t += stride, slot = slot + 1 == D ? 0 : slot + 1) {
        const unsigned tn = t + (D - 1) * stride;
        if (tn < nt) {
            issue(tn, slot == 0 ? D - 1 : slot - 1);
            if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
            else if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(57)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
        const bool in = v < V;
        const char *sl = (const char *)(ring + slot * SLOT);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NKI; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k], *(const float *)(sl + coff[k]), acc, 0, 0, 0);
        float *yb = y + (size_t)b * COUT * V + v + (h ? 4 * V : 0u);
#pragma unroll
        for (int r = 0; r < 12; ++r)
            if (in) yb[(size_t)((r & 3) + 8 * (r >> 2)) * V] = acc[r];
    }
}

template <int D, int NW>
static float run_dma16(const float *x, const float *W, float *y, int B, unsigned V, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (size_t)NW * D * ((CIN + 6) / 7) * 1024;
    CK(hipFuncSetAttribute((const void *)planar_dma16_kernel<D, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((planar_dma16_kernel<D, NW>), dim3(grid), dim3(64 * NW), lds, 0, x, W, y, B, V);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((planar_dma16_kernel<D, NW>), dim3(grid), dim3(64 * NW), lds, 0, x, W, y, B, V);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

// mode 8: the planar copy of mode 0 with NM MFMAs per tile that do NOT touch the copied data (own registers): does matrix-core
// activity by itself slow the streaming (clocks / issue), or is it the dependence chain load -> MFMA -> store?
template <int NM, int NW>
__global__ __launch_bounds__(64 * NW) void planar_copy_mfma_kernel(const float *__restrict__ x, float *__restrict__ y, int B, unsigned V, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const unsigned tpb = (V + 31) / 32, nt = tpb * B;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = seed;
    float wa = seed * lane, wb = seed + lane;
    for (unsigned t = blockIdx.x * NW + wave; t < nt; t += gridDim.x * NW) {
        const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
        const bool in = v < V;
        const float *xb = x + (size_t)b * CIN * V + (in ? v : 0u) + (h ? V : 0u);
        float xv[CIN / 2];
#pragma unroll
        for (int k = 0; k < CIN / 2; ++k) xv[k] = xb[(size_t)2 * k * V];
#pragma unroll
        for (int k = 0; k < NM; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, wb, acc, 0, 0, 0);
        float *yb = y + (size_t)b * COUT * V + v + (h ? V : 0u);
#pragma unroll
        for (int o = 0; o < COUT / 2; ++o) {
            const float s = xv[2 * o] + xv[2 * o + 1];
            if (in) yb[(size_t)2 * o * V] = s;
        }
    }
    float sacc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc += acc[r];
    if (sacc == 12345.678f) y[0] = sacc;
}

template <int NM, int NW>
static float run_copy_mfma(const float *x, float *y, int B, unsigned V, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((planar_copy_mfma_kernel<NM, NW>), dim3(grid), dim3(64 * NW), 0, 0, x, y, B, V, 0.5f);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((planar_copy_mfma_kernel<NM, NW>), dim3(grid), dim3(64 * NW), 0, 0, x, y, B, V, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

// mode 9: register double buffering, unrolled by two (no register copies): loads(t+1) -> MFMA(t) -> stores(t) -> loads(t+2) -> MFMA(t+1) -> ...
template <int NW>
__global__ __launch_bounds__(64 * NW) void planar_regpf_kernel(const float *__restrict__ x, const float *__restrict__ W, float *__restrict__ y, int B, unsigned V) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int NKI = CIN / 2;
    float w[NKI];
#pragma unroll
    for (int k = 0; k < NKI; ++k) w[k] = c < COUT ? W[c * CIN + 2 * k + h] : 0.f;
    const unsigned tpb = (V + 31) / 32, nt = tpb * B;
    const unsigned stride = gridDim.x * NW;
    auto load = [&](unsigned t, float (&xv)[NKI]) {
        const unsigned tt = t < nt ? t : nt - 1;
        const unsigned b = tt / tpb, v = (tt - b * tpb) * 32 + c;
        const float *xb = x + (size_t)b * CIN * V + (v < V ? v : 0u) + (h ? V : 0u);
#pragma unroll
        for (int k = 0; k < NKI; ++k) xv[k] = xb[(size_t)2 * k * V];
    };
    auto compute = [&](unsigned t, const float (&xv)[NKI]) {
        if (t >= nt) return;
        const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
        const bool in = v < V;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NKI; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k], xv[k], acc, 0, 0, 0);
        float *yb = y + (size_t)b * COUT * V + v + (h ? 4 * V : 0u);
#pragma unroll
        for (int r = 0; r < 12; ++r)
            if (in) yb[(size_t)((r & 3) + 8 * (r >> 2)) * V] = acc[r];
    };
    float xa[NKI], xb2[NKI];
    unsigned t = blockIdx.x * NW + wave;
    load(t, xa);
    for (; t < nt; t += 2 * stride) {
        load(t + stride, xb2);
        compute(t, xa);
        load(t + 2 * stride, xa);
        compute(t + stride, xb2);
    }
}

template <int NW>
static float run_regpf(const float *x, const float *W, float *y, int B, unsigned V, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((planar_regpf_kernel<NW>), dim3(grid), dim3(64 * NW), 0, 0, x, W, y, B, V);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((planar_regpf_kernel<NW>), dim3(grid), dim3(64 * NW), 0, 0, x, W, y, B, V);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

// mode 11: planar copy where every output depends on ALL loads of the tile (so the 12 stores leave as one burst after the last load
// has landed, as after an MFMA chain) -- against mode 0 where each store leaves as soon as its two inputs are there.
template <int NW>
__global__ __launch_bounds__(64 * NW) void planar_burst_kernel(const float *__restrict__ x, float *__restrict__ y, int B, unsigned V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const unsigned tpb = (V + 31) / 32, nt = tpb * B;
    for (unsigned t = blockIdx.x * NW + wave; t < nt; t += gridDim.x * NW) {
        const unsigned b = t / tpb, v = (t - b * tpb) * 32 + c;
        const bool in = v < V;
        const float *xb = x + (size_t)b * CIN * V + (in ? v : 0u) + (h ? V : 0u);
        float xv[CIN / 2];
#pragma unroll
        for (int k = 0; k < CIN / 2; ++k) xv[k] = xb[(size_t)2 * k * V];
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < CIN / 2; ++k) tot += xv[k];
        float *yb = y + (size_t)b * COUT * V + v + (h ? V : 0u);
#pragma unroll
        for (int o = 0; o < COUT / 2; ++o)
            if (in) yb[(size_t)2 * o * V] = tot + xv[o];
    }
}
template <int NW>
static float run_burst(const float *x, float *y, int B, unsigned V, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((planar_burst_kernel<NW>), dim3(grid), dim3(64 * NW), 0, 0, x, y, B, V);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((planar_burst_kernel<NW>), dim3(grid), dim3(64 * NW), 0, 0, x, y, B, V);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

template <int MODE, int NW>
static float run(const float *x, float *y, int B, unsigned V, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((planar_kernel<MODE, NW>), dim3(grid), dim3(64 * NW), 0, 0, x, y, B, V);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((planar_kernel<MODE, NW>), dim3(grid), dim3(64 * NW), 0, 0, x, y, B, V);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char **argv) {
    const int B = 2;
    for (int N : {65, 64}) {
        const unsigned V = (unsigned)N * N * N;
        float *x, *y;
        CK(hipMalloc(&x, (size_t)B * CIN * V * 4 + 64)); CK(hipMalloc(&y, (size_t)B * COUT * V * 4 + 64));
        CK(hipMemset(x, 0, (size_t)B * CIN * V * 4)); CK(hipMemset(y, 0, (size_t)B * COUT * V * 4));
        const double bytes = 4.0 * B * V * (CIN + COUT);
        for (int grid : {256}) {
            float t;
            t = run<0, 8>(x, y, B, V, grid, 20); printf("N=%d mode0 (2x128B/instr) nw8 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
            t = run<0, 4>(x, y, B, V, grid, 20); printf("N=%d mode0 (2x128B/instr) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
            t = run<1, 4>(x, y, B, V, grid, 20); printf("N=%d mode1 (1x256B/instr) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
            t = run<12, 8>(x, y, B, V, grid, 20); printf("N=%d mode12 (mode 0, contiguous tiles per wave) nw8 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
            t = run<12, 4>(x, y, B, V, grid, 20); printf("N=%d mode12 (mode 0, contiguous tiles per wave) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
            if (N % 4 == 0) { t = run<2, 4>(x, y, B, V, grid, 20); printf("N=%d mode2 (1x1KB aligned)  nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3); }
            t = run<3, 4>(x, y, B, V, grid, 20); printf("N=%d mode3 (1x1KB row-aligned) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
        }
        for (int grid : {256, 512, 1024}) {
            float t;
            t = run_burst<8>(x, y, B, V, grid, 20); printf("N=%d mode11 copy, stores after all loads nw8 grid %d: %6.1f us\n", N, grid, t);
            t = run_burst<4>(x, y, B, V, grid, 20); printf("N=%d mode11 copy, stores after all loads nw4 grid %d: %6.1f us\n", N, grid, t);
        }
        for (int grid : {256, 512}) {
            float t;
            t = run_copy_mfma<0, 8>(x, y, B, V, grid, 20); printf("N=%d mode8 copy +  0 independent MFMA nw8 grid %d: %6.1f us\n", N, grid, t);
            t = run_copy_mfma<12, 8>(x, y, B, V, grid, 20); printf("N=%d mode8 copy + 12 independent MFMA nw8 grid %d: %6.1f us\n", N, grid, t);
            t = run_copy_mfma<24, 8>(x, y, B, V, grid, 20); printf("N=%d mode8 copy + 24 independent MFMA nw8 grid %d: %6.1f us\n", N, grid, t);
            t = run_copy_mfma<48, 8>(x, y, B, V, grid, 20); printf("N=%d mode8 copy + 48 independent MFMA nw8 grid %d: %6.1f us\n", N, grid, t);
        }
        {
            float *W, *y2;
            CK(hipMalloc(&W, COUT * CIN * 4)); CK(hipMalloc(&y2, (size_t)B * COUT * V * 4 + 64));
            std::vector<float> hx((size_t)B * CIN * V), hw(COUT * CIN);
            for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.f - 1.f;
            for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u >> 3) & 255) / 256.f - .5f;
            CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
            run_mfma<4, 8>(x, W, y, B, V, 256, 1);
            run_mfma<6, 8>(x, W, y2, B, V, 256, 1);
            std::vector<float> a((size_t)B * COUT * V), bb(a.size());
            CK(hipMemcpy(a.data(), y, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bb.data(), y2, a.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; double mx = 0;
            for (size_t i = 0; i < a.size(); ++i) { if (a[i] != bb[i]) ++bad; if (fabs(a[i]) > mx) mx = fabs(a[i]); }
            printf("N=%d mode5 vs mode4: %zu of %zu differ (max |y| %.3f)\n", N, bad, a.size(), mx);
            CK(hipMemset(y2, 0, a.size() * 4));
            run_dma16<3, 4>(x, W, y2, B, V, 512, 1);
            CK(hipMemcpy(bb.data(), y2, a.size() * 4, hipMemcpyDeviceToHost));
            bad = 0;
            for (size_t i = 0; i < a.size(); ++i) if (a[i] != bb[i]) ++bad;
            printf("N=%d mode7 vs mode4: %zu of %zu differ\n", N, bad, a.size());
            for (int grid : {256}) {
                float t;
                t = run_dma16<2, 4>(x, W, y, B, V, grid, 20); printf("N=%d mode7 (16B DMA, D=2) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_dma16<3, 4>(x, W, y, B, V, grid, 20); printf("N=%d mode7 (16B DMA, D=3) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_dma16<4, 4>(x, W, y, B, V, grid, 20); printf("N=%d mode7 (16B DMA, D=4) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_dma16<2, 8>(x, W, y, B, V, grid, 20); printf("N=%d mode7 (16B DMA, D=2) nw8 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
            }
            CK(hipMemset(y2, 0, a.size() * 4));
            run_regpf<8>(x, W, y2, B, V, 256, 1);
            CK(hipMemcpy(bb.data(), y2, a.size() * 4, hipMemcpyDeviceToHost));
            bad = 0;
            for (size_t i = 0; i < a.size(); ++i) if (a[i] != bb[i]) ++bad;
            printf("N=%d mode9 vs mode4: %zu of %zu differ\n", N, bad, a.size());
            for (int grid : {256, 512, 1024}) {
                float t;
                t = run_regpf<4>(x, W, y, B, V, grid, 20); printf("N=%d mode9 (register double buffer) nw4 grid %4d: %6.1f us\n", N, grid, t);
                t = run_regpf<8>(x, W, y, B, V, grid, 20); printf("N=%d mode9 (register double buffer) nw8 grid %4d: %6.1f us\n", N, grid, t);
            }
            for (int abl : {1, 2, 3}) {
                float t = run_mfma<5, 4>(x, W, y2, B, V, 256, 20, abl); printf("N=%d mode5 nw4 grid 256 ablation %d (1 = no MFMA, 2 = no stores): %6.1f us\n", N, abl, t);
                t = run_mfma<5, 8>(x, W, y2, B, V, 256, 20, abl); printf("N=%d mode5 nw8 grid 256 ablation %d: %6.1f us\n", N, abl, t);
            }
            for (int grid : {256}) {
                float t;
                t = run_mfma<4, 8>(x, W, y, B, V, grid, 20); printf("N=%d mode4 (loads+MFMA+stores) nw8 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_mfma<4, 4>(x, W, y, B, V, grid, 20); printf("N=%d mode4 (loads+MFMA+stores) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_mfma<5, 8>(x, W, y, B, V, grid, 20); printf("N=%d mode5 (LDS-DMA ring, 1 ahead) nw8 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_mfma<5, 4>(x, W, y, B, V, grid, 20); printf("N=%d mode5 (LDS-DMA ring, 1 ahead) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_mfma<6, 8>(x, W, y, B, V, grid, 20); printf("N=%d mode6 (LDS-DMA ring, 2 ahead) nw8 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_mfma<6, 4>(x, W, y, B, V, grid, 20); printf("N=%d mode6 (LDS-DMA ring, 2 ahead) nw4 grid %4d: %6.1f us %5.0f GB/s\n", N, grid, t, bytes / t / 1e3);
                t = run_mfma<6, 2>(x, W, y, B, V, 2 * grid, 20); printf("N=%d mode6 (LDS-DMA ring, 2 ahead) nw2 grid %4d: %6.1f us %5.0f GB/s\n", N, 2 * grid, t, bytes / t / 1e3);
                t = run_mfma<5, 2>(x, W, y, B, V, 2 * grid, 20); printf("N=%d mode5 (LDS-DMA ring, 1 ahead) nw2 grid %4d: %6.1f us %5.0f GB/s\n", N, 2 * grid, t, bytes / t / 1e3);
            }
            CK(hipFree(W)); CK(hipFree(y2));
        }
        CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}
