// Development microbench: launch geometry of the 1-row GEMV stages of the depth loop (cache-resident weights, dependent
// chain captured in a hipGraph).  Same arithmetic structure as kernels_lm.hip::k_gemv (lane l owns 16-byte chunks l, l + 64, ...
// of x and of each weight row; optional RMSNorm prologue redone by every wave; butterfly; lane 0 stores), templated on threads
// per block and columns per wave; the grid follows.  Prints us per stage for the four depth-layer shapes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemv_geom_probe tools/gemv_geom_probe.hip && tools/bin/gemv_geom_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint16_t bf16_t;
__device__ __forceinline__ float bflo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bfhi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ bf16_t f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }
__device__ __forceinline__ float dot8(uint4 w, uint4 x, float s) {
    s = fmaf(bflo(w.x), bflo(x.x), s); s = fmaf(bfhi(w.x), bfhi(x.x), s); s = fmaf(bflo(w.y), bflo(x.y), s); s = fmaf(bfhi(w.y), bfhi(x.y), s);
    s = fmaf(bflo(w.z), bflo(x.z), s); s = fmaf(bfhi(w.z), bfhi(x.z), s); s = fmaf(bflo(w.w), bflo(x.w), s); s = fmaf(bfhi(w.w), bfhi(x.w), s);
    return s;
}
__device__ __forceinline__ float sq8(uint4 x, float s) { return dot8(x, x, s); }
__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}
__device__ __forceinline__ uint4 normc(uint4 v, uint4 g, float r) {
    uint4 o;
    o.x = pack2((bflo(v.x) * r) * bflo(g.x), (bfhi(v.x) * r) * bfhi(g.x)); o.y = pack2((bflo(v.y) * r) * bflo(g.y), (bfhi(v.y) * r) * bfhi(g.y));
    o.z = pack2((bflo(v.z) * r) * bflo(g.z), (bfhi(v.z) * r) * bfhi(g.z)); o.w = pack2((bflo(v.w) * r) * bflo(g.w), (bfhi(v.w) * r) * bfhi(g.w));
    return o;
}

template <int THR, int KC, int R, bool NORM>
__global__ __launch_bounds__(THR) void k_gv(const bf16_t* W, const bf16_t* x, const bf16_t* nw, bf16_t* y, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * (THR / 64) + wave) * R;
    if (n0 >= N) return;
    uint4 xv[KC], g[KC], w[R][KC];
    const uint4* xr = reinterpret_cast<const uint4*>(x);
#pragma unroll
    for (int j = 0; j < KC; ++j) xv[j] = xr[lane + 64 * j];
    if (NORM) {
#pragma unroll
        for (int j = 0; j < KC; ++j) g[j] = reinterpret_cast<const uint4*>(nw)[lane + 64 * j];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint4* wr = reinterpret_cast<const uint4*>(W + (size_t)(n0 + r) * KC * 512);
#pragma unroll
        for (int j = 0; j < KC; ++j) w[r][j] = wr[lane + 64 * j];
    }
    if (NORM) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < KC; ++j) s = sq8(xv[j], s);
        s = wave_sum(s);
        const float rinv = 1.0f / sqrtf(s / (float)(KC * 512) + 1e-6f);
#pragma unroll
        for (int j = 0; j < KC; ++j) xv[j] = normc(xv[j], g[j], rinv);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < KC; ++j) acc = dot8(w[r][j], xv[j], acc);
        acc = wave_sum(acc);
        if (lane == r) y[n0 + r] = f2bf(acc * 0.02f);
    }
}

static const int CHAIN = 120;
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

static bf16_t *W, *xa, *xb, *nw;
static const size_t WBYTES = (size_t)160 << 20;      // 160 MB of weights cycling: Infinity-Cache resident like the depth stack

template <int THR, int KC, int R, bool NORM>
static void run(hipStream_t st, int N, const char* tag) {
    const int K = KC * 512;
    const size_t per = (size_t)N * K, slots = WBYTES / 2 / per;
    const int waves = N / R, grid = (waves + THR / 64 - 1) / (THR / 64);
    const float us = time_chain(st, [&](int i) {
        hipLaunchKernelGGL((k_gv<THR, KC, R, NORM>), dim3(grid), dim3(THR), 0, st, W + (i % slots) * per, (i & 1) ? xa : xb, nw, (i & 1) ? xb : xa, N);
    });
    printf("  %-10s N %5d K %5d %s  block %4d x %d col/wave  grid %4d : %6.2f us\n", tag, N, K, NORM ? "norm" : "copy", THR, R, grid, us);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(hipMalloc(&W, WBYTES)); CK(hipMalloc(&xa, 8192 * 2)); CK(hipMalloc(&xb, 8192 * 2)); CK(hipMalloc(&nw, 8192 * 2));
    {
        std::vector<bf16_t> h(WBYTES / 2);
        uint32_t r = 1u;
        for (size_t i = 0; i < h.size(); ++i) { r = r * 1664525u + 1013904223u; h[i] = (bf16_t)(((r >> 16) & 0x80ffu) | 0x3c00u); }
        CK(hipMemcpy(W, h.data(), WBYTES, hipMemcpyHostToDevice));
        CK(hipMemcpy(xa, h.data(), 8192 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(xb, h.data() + 9000, 8192 * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(nw, h.data() + 20000, 8192 * 2, hipMemcpyHostToDevice));
    }
    printf("qkv: N 4096, K 1024, norm prologue (engine: 256 thr x 2 col/wave, grid 512: 4.7-4.8 us)\n");
    run<256, 2, 2, true>(st, 4096, "engine");
    run<256, 2, 1, true>(st, 4096, "");
    run<512, 2, 2, true>(st, 4096, "");
    run<512, 2, 1, true>(st, 4096, "");
    run<1024, 2, 2, true>(st, 4096, "");
    run<1024, 2, 1, true>(st, 4096, "");
    run<512, 2, 4, true>(st, 4096, "");
    run<256, 2, 4, true>(st, 4096, "");
    run<512, 2, 2, false>(st, 4096, "no norm");
    printf("gate/up as N 6144 (engine: 256 thr x (1 gate + 1 up), grid 768: 4.8-5.2 us)\n");
    run<256, 2, 2, true>(st, 6144, "engine");
    run<512, 2, 2, true>(st, 6144, "");
    run<1024, 2, 2, true>(st, 6144, "");
    run<512, 2, 4, true>(st, 6144, "");
    run<1024, 2, 4, true>(st, 6144, "");
    run<256, 2, 6, true>(st, 6144, "");
    run<512, 2, 6, true>(st, 6144, "");
    printf("down: N 1024, K 3072, copy (engine: 256 thr x 1 col/wave, grid 256: 4.6 us)\n");
    run<256, 6, 1, false>(st, 1024, "engine");
    run<512, 6, 1, false>(st, 1024, "");
    run<1024, 6, 1, false>(st, 1024, "");
    run<128, 6, 1, false>(st, 1024, "");
    run<64, 6, 1, false>(st, 1024, "");
    printf("o_proj: N 1024, K 2048, copy\n");
    run<256, 4, 1, false>(st, 1024, "");
    run<512, 4, 1, false>(st, 1024, "");
    run<128, 4, 1, false>(st, 1024, "");
    run<64, 4, 1, false>(st, 1024, "");
    printf("head: N 2048, K 1024, norm\n");
    run<256, 2, 1, true>(st, 2048, "engine");
    run<512, 2, 1, true>(st, 2048, "");
    run<256, 2, 2, true>(st, 2048, "");
    run<128, 2, 1, true>(st, 2048, "");
    return 0;
}
