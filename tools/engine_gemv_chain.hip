// Development microbench: the ENGINE's own 1-row GEMV kernels (kernels_lm.hip, included verbatim) as a dependent hipGraph chain
// over cache-resident weights, next to tools/gemv_geom_probe.hip's minimal kernel of the same geometry: is the 0.9-1.5 us per
// depth stage that the frame pays above the minimal chain a property of the kernels or of the frame around them?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vox_serve_amd/csrc -o tools/bin/engine_gemv_chain tools/engine_gemv_chain.hip
#include "../vox_serve_amd/csrc/kernels_lm.hip"
#include <stdarg.h>
#include <vector>
int vox_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return code; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static int CHAIN = 120;
template <typename F>
static float time_chain(hipStream_t st, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < CHAIN; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1000.f / (20 * CHAIN);
}
int main(int argc, char** argv) {
    if (argc > 1) CHAIN = atoi(argv[1]);
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t WB = (size_t)(argc > 2 ? atoi(argv[2]) : 160) << 20;      // weight footprint cycled by the chains (MB)
    bf16_t *W, *xa, *xb, *nw, *res;
    CK(hipMalloc(&W, WB)); CK(hipMalloc(&xa, 8192 * 2)); CK(hipMalloc(&xb, 8192 * 2)); CK(hipMalloc(&nw, 8192 * 2)); CK(hipMalloc(&res, 8192 * 2));
    {
        std::vector<bf16_t> h(WB / 2);
        uint32_t r = 1u;
        for (size_t i = 0; i < h.size(); ++i) { r = r * 1664525u + 1013904223u; h[i] = (bf16_t)(((r >> 16) & 0x80ffu) | 0x3c00u); }
        CK(hipMemcpy(W, h.data(), WB, hipMemcpyHostToDevice));
        CK(hipMemcpy(xa, h.data(), 8192 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(xb, h.data() + 9000, 8192 * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(nw, h.data() + 20000, 8192 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(res, h.data() + 30000, 8192 * 2, hipMemcpyHostToDevice));
    }
    vox_ctx ctx{}; ctx.n_cu = 256; ctx.exact_rows = 2;
    auto run = [&](const char* tag, int N, int K, int pro, int epi, bool residual, int keep, bool two) {
        const size_t per = (size_t)N * K * (two ? 2 : 1), slots = WB / 2 / per;
        const float us = time_chain(st, [&](int i) {
            LinearCall c;
            c.W = W + (i % slots) * per; if (two) c.W2 = (const bf16_t*)c.W + (size_t)N * K;
            c.x = (i & 1) ? xa : xb; c.y = (i & 1) ? xb : xa; c.B = 1; c.N = N; c.K = K; c.pro = pro; c.epi = epi; c.norm_w = nw; c.eps = 1e-6f;
            c.residual = residual ? res : nullptr; c.keep_weights = keep; c.fixed_order = 1;
            if (vox_launch_linear(&ctx, st, c) != VOX_OK) exit(1);
        });
        printf("  %-34s N %5d K %5d  %6.2f us per stage\n", tag, N, K, us);
    };
    printf("engine kernels (vox_launch_linear, 1 row, keep = cache-resident plain loads) as a dependent graph chain:\n");
    run("qkv: norm prologue", 4096, 1024, VOX_PRO_RMSNORM, VOX_EPI_STORE, false, 1, false);
    run("qkv: norm prologue, nt loads", 4096, 1024, VOX_PRO_RMSNORM, VOX_EPI_STORE, false, 0, false);
    run("gate/up: norm + SiLU*up", 3072, 1024, VOX_PRO_RMSNORM, VOX_EPI_SILU_MUL, false, 1, true);
    run("down: copy + residual", 1024, 3072, VOX_PRO_COPY, VOX_EPI_STORE, true, 1, false);
    run("down: copy, no residual", 1024, 3072, VOX_PRO_COPY, VOX_EPI_STORE, false, 1, false);
    run("o_proj: copy + residual", 1024, 2048, VOX_PRO_COPY, VOX_EPI_STORE, true, 1, false);
    run("head: norm", 2048, 1024, VOX_PRO_RMSNORM, VOX_EPI_STORE, false, 1, false);
    {   // one depth layer's four linears in sequence, own buffers per hand-off, 30 layers per graph
        bf16_t *qkv, *att, *h;
        CK(hipMalloc(&qkv, 8192 * 2)); CK(hipMalloc(&att, 8192 * 2)); CK(hipMalloc(&h, 8192 * 2));
        CK(hipMemcpy(qkv, xa, 8192 * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(att, xb, 8192 * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(h, xa, 8192 * 2, hipMemcpyDeviceToDevice));
        const size_t layer = (size_t)(4096 + 1024 * 2 + 3072 * 2 + 3072) * 1024;       // elements per layer
        const size_t slots = WB / 2 / layer;
        const float us = time_chain(st, [&](int i) {
            const bf16_t* base = W + ((size_t)(i / 4) % slots) * layer;
            LinearCall c;
            c.B = 1; c.eps = 1e-6f; c.keep_weights = 1; c.fixed_order = 1; c.norm_w = nw;
            switch (i & 3) {
            case 0: c.W = base; c.x = xa; c.y = qkv; c.N = 4096; c.K = 1024; c.pro = VOX_PRO_RMSNORM; c.epi = VOX_EPI_STORE; break;
            case 1: c.W = base + (size_t)4096 * 1024; c.x = att; c.y = xa; c.residual = xa; c.N = 1024; c.K = 2048; c.pro = VOX_PRO_COPY; c.epi = VOX_EPI_STORE; break;
            case 2: c.W = base + (size_t)6144 * 1024; c.W2 = (const bf16_t*)c.W + (size_t)3072 * 1024; c.x = xa; c.y = h; c.N = 3072; c.K = 1024; c.pro = VOX_PRO_RMSNORM; c.epi = VOX_EPI_SILU_MUL; break;
            default: c.W = base + (size_t)12288 * 1024; c.x = h; c.y = xa; c.residual = xa; c.N = 1024; c.K = 3072; c.pro = VOX_PRO_COPY; c.epi = VOX_EPI_STORE; break;
            }
            if (vox_launch_linear(&ctx, st, c) != VOX_OK) exit(1);
        });
        printf("  one layer's four linears in sequence (qkv, o_proj, gate/up, down; attention left out): %6.2f us per stage = %6.2f us per layer\n", us, 4 * us);
    }
    return 0;
}
