// Calibration: cycles per v_mfma_f32_16x16x4_f32 for one wave / two waves per SIMD, with and without VALU work between the MFMAs,
// and cycles per ds_read2_b32 for a single wave stream.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate.bin mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CHAINS, int VALU_PER, int DISTINCT>
__global__ __launch_bounds__(512) void k_mfma(float *out, long long *cyc, int iters) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[8], b = threadIdx.x * 0.001f, v = threadIdx.x;
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float av;
            if (DISTINCT) av = a[u];          // operand registers prepared ahead
            else { a[0] = a[0] + v; av = a[0]; }   // operand written right before the MFMA (same register every time)
            acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[u % CHAINS], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < VALU_PER; ++j) v = fmaf(v, 1.0001f, 0.5f);
        }
        if (DISTINCT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += v;
        }
    }
    long long t1 = clock64();
    float s = v;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ __launch_bounds__(512) void k_lds(float *out, long long *cyc, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const float *p = lds + (threadIdx.x & 63) * 65 % 4096;
    float s = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v[2 * u] = p[4 * u + (it & 1)];
            v[2 * u + 1] = p[4 * u + 130 + (it & 1)];
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
static int run(const char *name, K kern, int threads, int iters, int per_iter) {
    float *out; long long *cyc, h;
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 8));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters); }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-64s %3d threads/CU: %6.1f cycles per op\n", name, threads, (double)h / iters / per_iter);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    const int it = 2000;
    for (int threads : {256, 512}) {
        run("mfma 16x16x4 f32, 1 chain, operands ready", k_mfma<1, 0, 1>, threads, it, 8);
        run("mfma 16x16x4 f32, 2 chains, operands ready", k_mfma<2, 0, 1>, threads, it, 8);
        run("mfma 16x16x4 f32, 4 chains, operands ready", k_mfma<4, 0, 1>, threads, it, 8);
        run("mfma, 2 chains, operand = same register written before each", k_mfma<2, 0, 0>, threads, it, 8);
        run("mfma, 4 chains, operand = same register written before each", k_mfma<4, 0, 0>, threads, it, 8);
        run("mfma, 4 chains, ready operands + 2 VALU between", k_mfma<4, 2, 1>, threads, it, 8);
        run("mfma, 4 chains, ready operands + 4 VALU between", k_mfma<4, 4, 1>, threads, it, 8);
        run("mfma, 4 chains, ready operands + 6 VALU between", k_mfma<4, 6, 1>, threads, it, 8);
        run("ds_read2_b32-ish stream (32 dwords / iter), per dword", k_lds, threads, it, 32);
    }
    return 0;
}
