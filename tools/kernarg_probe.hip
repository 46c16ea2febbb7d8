// Development microbench: does kernarg preloading (user SGPRs filled by the CP: -mllvm -amdgpu-kernarg-preload-count=16, flat
// scalar arguments only — a by-value struct is never preloaded) shorten a dependent launch chain on MI355X?
// Every stage does what the engine's small stages do first: read pointers from the kernel arguments, load 16 bytes per lane
// through them (the previous stage's output), a little arithmetic, store.  Same source built twice:
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/kernarg_probe_off tools/kernarg_probe.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o tools/bin/kernarg_probe_on tools/kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
struct Args {      // shaped like the engine's LinArgs: pointers first, scalars after, ~200 bytes
    const float4 *x, *w, *w2, *bias, *res, *nw;
    float4 *y, *xo;
    const float *po, *pm;
    const int *kv, *rows;
    long xs, xos;
    float eps;
    int B, N, K, Hq, D, mc, keep, rt, yr;
    const float4 *wf, *w2f, *xf;
    float4* yf;
};
__global__ __launch_bounds__(512) void k_struct(Args a) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    float4 v = a.x[i], w = a.w[i & 4095];
    v.x = v.x * 0.999f + w.x * a.eps; v.y = v.y * 0.999f + w.y; v.z += w.z * 0.001f; v.w += w.w * 0.001f;
    a.y[i] = v;
}
__global__ __launch_bounds__(512) void k_flat(const float4* x, const float4* w, float4* y, float eps, int n) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    float4 v = x[i], ww = w[i & 4095];
    v.x = v.x * 0.999f + ww.x * eps; v.y = v.y * 0.999f + ww.y; v.z += ww.z * 0.001f; v.w += ww.w * 0.001f;
    y[i] = v;
}
int main() {
    const int grid = 256, n = grid * 512;
    float4 *b0, *b1, *w;
    CK(hipMalloc(&b0, n * 16)); CK(hipMalloc(&b1, n * 16)); CK(hipMalloc(&w, 4096 * 16));
    CK(hipMemset(b0, 0, n * 16)); CK(hipMemset(b1, 0, n * 16)); CK(hipMemset(w, 0, 4096 * 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        auto body = [&]() {
            for (int i = 0; i < N; ++i) {
                float4 *in = (i & 1) ? b1 : b0, *out = (i & 1) ? b0 : b1;
                if (mode == 0) { Args a{}; a.x = in; a.w = w; a.y = out; a.eps = 1e-3f; a.B = n; hipLaunchKernelGGL(k_struct, dim3(grid), dim3(512), 0, st, a); }
                else hipLaunchKernelGGL(k_flat, dim3(grid), dim3(512), 0, st, (const float4*)in, (const float4*)w, out, 1e-3f, n);
            }
        };
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); body(); CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-8s graph chain  %6.3f us/stage\n", mode ? "flat" : "struct", ms * 1000 / N);
        }
    }
    return 0;
}
