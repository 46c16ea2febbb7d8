// Development: time vox_linear through the C ABI without Python in the loop.  hipcc -O2 -o bin/linear_bench linear_bench.cpp -L../vox_serve_amd -lvoxhip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <dlfcn.h>
#include "../include/voxhip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char** argv) {
    void* h = dlopen(argv[1], RTLD_NOW); if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    auto ctx_create = (int (*)(int, vox_ctx**))dlsym(h, "vox_ctx_create");
    auto linear = (int (*)(vox_ctx*, void*, const void*, const void*, const void*, const void*, void*, int, int, int, int))dlsym(h, "vox_linear");
    auto gb = (int (*)(vox_ctx*, void*))dlsym(h, "vox_graph_begin");
    auto ge = (int (*)(vox_ctx*, void*, vox_graph**))dlsym(h, "vox_graph_end");
    auto gl = (int (*)(vox_graph*, void*))dlsym(h, "vox_graph_launch");
    vox_ctx* ctx; ctx_create(0, &ctx);
    hipStream_t st; CK(hipStreamCreate(&st));
    int shapes[][3] = {{32,1024,2048},{32,1024,3072},{32,4096,1024},{32,4096,2048},{32,12288,2048},{32,2048,6144},{16,1024,2048},{1,1024,2048},{1,4096,2048},{8,1024,2048},{75,4096,2048}};
    const int NB = 24; size_t maxb = (size_t)12288 * 2048 * 2;
    char* W; CK(hipMalloc(&W, maxb * NB)); CK(hipMemset(W, 0x11, maxb * NB));
    void *x, *y, *res; CK(hipMalloc(&x, 1 << 22)); CK(hipMalloc(&y, 1 << 23)); CK(hipMalloc(&res, 1 << 23)); CK(hipMemset(x, 0x11, 1 << 22)); CK(hipMemset(res, 0, 1 << 23));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto& s : shapes) {
        const int B = s[0], N = s[1], K = s[2], IT = 200;
        for (int i = 0; i < 3; ++i) linear(ctx, st, W + (size_t)(i % NB) * maxb, nullptr, x, res, y, B, N, K, 0);
        CK(hipStreamSynchronize(st));
        vox_graph* g; gb(ctx, st);
        for (int i = 0; i < IT; ++i) linear(ctx, st, W + (size_t)(i % NB) * maxb, nullptr, x, res, y, B, N, K, 0);
        ge(ctx, st, &g);
        gl(g, st); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st)); gl(g, st); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("B=%3d N=%5d K=%5d  %7.2f us  %5.2f TB/s\n", B, N, K, ms * 1000 / IT, (double)N * K * 2 / (ms * 1000 / IT) / 1e6);
    }
    return 0;
}
