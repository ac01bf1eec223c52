// Per-kernel floor of a dependent chain: N tiny kernels back to back on a stream, and as a captured graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void tiny(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
    float *p; CK(hipMalloc(&p, 1 << 24)); CK(hipMemset(p, 0, 1 << 24));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 200;
    for (int blocks : {1, 256, 2352, 16384}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(64), 0, s, p, blocks * 64);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("stream: %5d blocks x 64 threads: %.2f us per kernel\n", blocks, ms * 1e3 / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(64), 0, s, p, blocks * 64);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("graph : %5d blocks x 64 threads: %.2f us per kernel\n", blocks, ms * 1e3 / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
