// Development microbench: cost of a dependent kernel boundary, eager vs hipGraph, on this box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_chain(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] + 1; }
__global__ void k_wide(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
int main() {
    int* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    float* f; CK(hipMalloc(&f, 1 << 22));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 2000;
    auto run = [&](const char* name, int mode, int grid) {
        auto body = [&]() {
            for (int i = 0; i < N; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, d);
                else if (mode == 1) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, st, d);
                else hipLaunchKernelGGL(k_wide, dim3(grid), dim3(256), 0, st, f, grid * 256);
            }
        };
        body(); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st)); body(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s grid %4d eager  %6.2f us/kernel\n", name, grid, ms * 1000 / N);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); body(); CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s grid %4d graph  %6.2f us/kernel\n", name, grid, ms * 1000 / N);
    };
    run("empty", 0, 1); run("empty", 0, 256); run("chain rmw", 1, 1); run("wide rmw", 2, 256); run("wide rmw", 2, 1024);
    return 0;
}
