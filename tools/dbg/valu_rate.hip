// Calibration: wave-instruction cost of plain / packed fp32 VALU ops and v_exp_f32 at 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters) {
    float v[8];
    f32x2 w[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i; w[i] = f32x2{v[i], v[i] + 0.5f}; }
    const float c = 1.0001f, d = 0.37f;
    const f32x2 c2 = {c, c}, d2 = {d, d};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) v[u] = fmaf(v[u], c, d);
            if (MODE == 1) w[u] = __builtin_elementwise_fma(w[u], c2, d2);
            if (MODE == 2) v[u] = __builtin_amdgcn_exp2f(v[u]);
            if (MODE == 3) v[u] = v[u] > 0.5f ? v[u] * c : d;
            if (MODE == 4) w[u] = w[u] * c2;
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + w[i][0] + w[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <typename K>
static int run(const char *name, K kern, int threads, int iters) {
    float *out; long long *cyc, h;
    CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&cyc, 8));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-36s %4d threads/CU (%d waves/SIMD): %6.2f cycles per instruction per wave, %5.2f per SIMD\n", name, threads, threads / 256,
           (double)h / iters / 8, (double)h / iters / 8 / (threads / 256));
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}
int main() {
    const int it = 4000;
    for (int threads : {256, 512, 768, 1024}) {
        run("v_fma_f32", k<0>, threads, it);
        run("v_pk_fma_f32", k<1>, threads, it);
        run("v_pk_mul_f32", k<4>, threads, it);
        run("v_exp_f32", k<2>, threads, it);
        run("cmp + mul + cndmask (3 instr)", k<3>, threads, it);
    }
    return 0;
}
