// Micro-benchmark: streaming read of C channel rows (odd row length) with 4-, 8- and 16-byte per-lane loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) f2u { float x, y; };

template <int VEC, int CH>
__global__ __launch_bounds__(256) void rd(const float* __restrict__ x, float* out, unsigned V, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const unsigned tv = 32 * VEC;
    const unsigned tiles_per_b = (V + tv - 1) / tv, ntiles = tiles_per_b * B;
    float acc = 0.f;
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
        const unsigned b = t / tiles_per_b;
        unsigned v = (t - b * tiles_per_b) * tv + c * VEC;
        if (v + VEC > V) v = 0;
        const float* xb = x + (size_t)b * CH * V;
        float r[CH / 2][VEC];
#pragma unroll
        for (int ks = 0; ks < CH / 2; ++ks) {
            const float* p = xb + (size_t)(2 * ks) * V + (h ? V : 0u) + v;
            if (VEC == 4) { f4u q = *(const f4u*)p; r[ks][0] = q.x; r[ks][1 % VEC] = q.y; r[ks][2 % VEC] = q.z; r[ks][3 % VEC] = q.w; }
            else if (VEC == 2) { f2u q = *(const f2u*)p; r[ks][0] = q.x; r[ks][1 % VEC] = q.y; }
            else r[ks][0] = *p;
        }
#pragma unroll
        for (int ks = 0; ks < CH / 2; ++ks)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc += r[ks][j];
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int VEC, int CH>
float run(const float* x, float* out, unsigned V, int B, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rd<VEC, CH>), dim3(grid), dim3(256), 0, 0, x, out, V, B);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((rd<VEC, CH>), dim3(grid), dim3(256), 0, 0, x, out, V, B);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}
int main() {
    const unsigned V = 65 * 65 * 65; const int B = 2, CH = 48;
    size_t n = (size_t)B * CH * V + 64;
    float *x, *out; hipMalloc(&x, n * 4); hipMalloc(&out, 64); hipMemset(x, 0, n * 4);
    const double mb = (double)B * CH * V * 4 / 1e6;
    for (int grid : {1024, 2048, 4096}) {
        float t1 = run<1, CH>(x, out, V, B, grid), t2 = run<2, CH>(x, out, V, B, grid), t4 = run<4, CH>(x, out, V, B, grid);
        printf("grid %d: dword %.1f us (%.0f GB/s)  dwordx2 %.1f us (%.0f GB/s)  dwordx4 %.1f us (%.0f GB/s)\n", grid, t1, mb / t1 * 1e3, t2, mb / t2 * 1e3, t4, mb / t4 * 1e3);
    }
    return 0;
}
