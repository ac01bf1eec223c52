// Does a planar streaming kernel gain from bf16 rows when its tiles stay 32 voxels wide (64-byte row segments), or does it need
// 64-voxel tiles (128-byte segments)?  48 input rows, 24 output rows per sample, 2 x 65^3 voxels (channel stride padded to 64 elements).
//   mode 0: fp32 rows, 32-voxel tiles (2 rows x 128 B per wave instruction)          -- the shipped pointwise kernels
//   mode 1: bf16 rows, 32-voxel tiles (2 rows x 64 B per wave instruction, 2-byte accesses)
//   mode 2: bf16 rows, 64-voxel tiles (2 rows x 128 B per wave instruction, 4-byte accesses = two voxels per lane)
//   mode 3: half the rows fp32 and half bf16, 32-voxel tiles   (the FNOSeg block tail under autocast: s, y fp32; x, out bf16)
//   mode 4: the same mix, 64-voxel tiles (fp32 rows as 8-byte accesses)
// Build: hipcc --offload-arch=gfx950 -O3 -o seg_bench seg_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int CIN = 48, COUT = 24, NW = 8;
typedef unsigned short u16;

template <int MODE>
__global__ __launch_bounds__(64 * NW) void k(const char *__restrict__ x, char *__restrict__ y, int B, unsigned V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
    constexpr int TV = (MODE == 2 || MODE == 4) ? 64 : 32;
    const unsigned tpb = V / TV, nt = tpb * B;
    for (unsigned t = blockIdx.x * NW + wave; t < nt; t += gridDim.x * NW) {
        const unsigned b = t / tpb, v0 = (t - b * tpb) * TV;
        float acc[COUT / 2];
#pragma unroll
        for (int o = 0; o < COUT / 2; ++o) acc[o] = 0.f;
#pragma unroll
        for (int kk = 0; kk < CIN / 2; ++kk) {
            const size_t row = (size_t)b * CIN + 2 * kk + h;
            const bool half = (MODE >= 3) && (kk >= CIN / 4);      // second half of the rows is bf16 in the mixed modes
            float val;
            if (MODE == 0 || (MODE == 3 && !half)) val = ((const float *)x)[row * V + v0 + c];
            else if (MODE == 1 || (MODE == 3 && half)) val = (float)((const u16 *)x)[row * V + v0 + c];
            else if (MODE == 2 || (MODE == 4 && half)) { unsigned w = ((const unsigned *)x)[(row * V + v0) / 2 + c]; val = (float)(w & 0xffff) + (float)(w >> 16); }
            else { float2 w = ((const float2 *)x)[(row * V + v0) / 2 + c]; val = w.x + w.y; }
            acc[kk % (COUT / 2)] += val;
        }
#pragma unroll
        for (int o = 0; o < COUT / 2; ++o) {
            const size_t row = (size_t)b * COUT + 2 * o + h;
            const bool half = (MODE >= 3) && (o >= COUT / 4);
            if (MODE == 0 || (MODE == 3 && !half)) ((float *)y)[row * V + v0 + c] = acc[o];
            else if (MODE == 1 || (MODE == 3 && half)) ((u16 *)y)[row * V + v0 + c] = (u16)acc[o];
            else if (MODE == 2 || (MODE == 4 && half)) ((unsigned *)y)[(row * V + v0) / 2 + c] = (unsigned)acc[o];
            else ((float2 *)y)[(row * V + v0) / 2 + c] = float2{acc[o], acc[o]};
        }
    }
}

template <int MODE>
static void run(const char *x, char *y, int B, unsigned V, double bytes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256, 512}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * NW), 0, 0, x, y, B, V);
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * NW), 0, 0, x, y, B, V);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d grid %d: %7.2f us  %6.0f GB/s (%.1f MB)\n", MODE, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9, bytes / 1e6);
    }
}

int main() {
    const int B = getenv("SEG_B") ? atoi(getenv("SEG_B")) : 2; const unsigned V = 274688;     // 65^3 rounded up to 64
    char *x, *y;
    CK(hipMalloc(&x, (size_t)B * CIN * V * 4)); CK(hipMalloc(&y, (size_t)B * COUT * V * 4));
    CK(hipMemset(x, 0, (size_t)B * CIN * V * 4));
    const double e = (double)B * (CIN + COUT) * V;
    run<0>(x, y, B, V, e * 4); run<1>(x, y, B, V, e * 2); run<2>(x, y, B, V, e * 2); run<3>(x, y, B, V, e * 3); run<4>(x, y, B, V, e * 3);
    return 0;
}
