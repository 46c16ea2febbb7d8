// Development microbench: does a stage that ALSO requests the next stage's weights (same block id = same XCD, so they land in
// the L2 the next kernel's block will read from) shorten a chain of dependent GEMV stages?
// Chain of CHAIN dependent wave-per-2-rows GEMVs y = W_s x (N x K bf16), one hipGraph, weights rotating over NMAT matrices.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/prefetch_probe tools/prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint16_t bf16_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ bf16_t f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
__device__ __forceinline__ float dot8(uint4 w, uint4 x, float s) {
    s = fmaf(bflo(w.x), bflo(x.x), s); s = fmaf(bfhi(w.x), bfhi(x.x), s);
    s = fmaf(bflo(w.y), bflo(x.y), s); s = fmaf(bfhi(w.y), bfhi(x.y), s);
    s = fmaf(bflo(w.z), bflo(x.z), s); s = fmaf(bfhi(w.z), bfhi(x.z), s);
    s = fmaf(bflo(w.w), bflo(x.w), s); s = fmaf(bfhi(w.w), bfhi(x.w), s);
    return s;
}
__device__ __forceinline__ uint4 ldg_nt(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// K = 512*KC; a wave owns R consecutive rows; block = 4 waves.  PF: also request the same rows of Wnext (discarded).
template <int KC, int R, int PF, int NT>
__global__ __launch_bounds__(256) void k_stage(const bf16_t* W, const bf16_t* Wnext, const bf16_t* x, bf16_t* y, int N, int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = 512 * KC;
    const int n0 = (blockIdx.x * 4 + wave) * R;
    uint4 xv[KC], w[R][KC], pf[R][KC];
    const uint4* xr = reinterpret_cast<const uint4*>(x);
#pragma unroll
    for (int j = 0; j < KC; ++j) xv[j] = xr[lane + 64 * j];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint4* wr = reinterpret_cast<const uint4*>(W + (size_t)(n0 + r) * K);
#pragma unroll
        for (int j = 0; j < KC; ++j) w[r][j] = NT ? ldg_nt(wr + lane + 64 * j) : wr[lane + 64 * j];
    }
    if (PF) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4* wr = reinterpret_cast<const uint4*>(Wnext + (size_t)(n0 + r) * K);
#pragma unroll
            for (int j = 0; j < KC; ++j) pf[r][j] = wr[lane + 64 * j];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < KC; ++j) s = dot8(w[r][j], xv[j], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) y[n0 + r] = f2bf(s * 0.01f);
    }
    if (PF) {
        uint32_t d = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < KC; ++j) d ^= pf[r][j].x ^ pf[r][j].y ^ pf[r][j].z ^ pf[r][j].w;
        if (d == 0x1234567u) sink[0] = (int)d;       // never true for this data; keeps the loads alive
    }
}

static const int CHAIN = 60;
template <typename F>
static float time_chain(hipStream_t st, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < CHAIN; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int reps = 30;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1000.f / (reps * CHAIN);
}

template <int KC, int R>
static void shape(hipStream_t st, int N, std::vector<bf16_t*>& Ws, bf16_t* xa, bf16_t* xb, int* sink) {
    const int K = 512 * KC, grid = N / (4 * R);
    printf("N=%d K=%d (%.1f MB per stage), grid %d x 256, %d rows per wave\n", N, K, (double)N * K * 2 / 1e6, grid, R);
    for (int nmat : {1, 2, 4, 10, 20, 60}) {
        if ((size_t)nmat > Ws.size()) continue;
        auto run = [&](int pf, int nt) {
            return time_chain(st, [&](int i) {
                const bf16_t* W = Ws[i % nmat]; const bf16_t* Wn = Ws[(i + 1) % nmat];
                bf16_t* xi = (i & 1) ? xb : xa; bf16_t* yo = (i & 1) ? xa : xb;
                if (pf && nt) hipLaunchKernelGGL((k_stage<KC, R, 1, 1>), dim3(grid), dim3(256), 0, st, W, Wn, xi, yo, N, sink);
                else if (pf) hipLaunchKernelGGL((k_stage<KC, R, 1, 0>), dim3(grid), dim3(256), 0, st, W, Wn, xi, yo, N, sink);
                else if (nt) hipLaunchKernelGGL((k_stage<KC, R, 0, 1>), dim3(grid), dim3(256), 0, st, W, Wn, xi, yo, N, sink);
                else hipLaunchKernelGGL((k_stage<KC, R, 0, 0>), dim3(grid), dim3(256), 0, st, W, Wn, xi, yo, N, sink);
            });
        };
        const float a = run(0, 0), b = run(1, 0), c = run(0, 1), d = run(1, 1);
        printf("  %2d matrices (%6.1f MB): plain %5.2f us | + prefetch next %5.2f us | nt %5.2f us | nt + prefetch(plain) %5.2f us\n",
               nmat, (double)nmat * N * K * 2 / 1e6, a, b, c, d);
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t maxW = (size_t)6144 * 2048;      // 25 MB
    std::vector<bf16_t*> Ws;
    std::vector<bf16_t> h(maxW);
    uint32_t r = 12345u;
    for (size_t i = 0; i < maxW; ++i) { r = r * 1664525u + 1013904223u; h[i] = (bf16_t)(((r >> 16) & 0x80ffu) | 0x3c00u | ((r >> 9) & 0x0300u)); }
    for (int i = 0; i < 60; ++i) { bf16_t* p; CK(hipMalloc(&p, maxW * 2)); CK(hipMemcpy(p, h.data(), maxW * 2, hipMemcpyHostToDevice)); Ws.push_back(p); }
    bf16_t *xa, *xb; int* sink;
    CK(hipMalloc(&xa, 65536)); CK(hipMalloc(&xb, 65536)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(xa, h.data(), 65536, hipMemcpyHostToDevice)); CK(hipMemcpy(xb, h.data() + 999, 65536, hipMemcpyHostToDevice));
    shape<2, 2>(st, 4096, Ws, xa, xb, sink);       // depth qkv: 8.4 MB
    shape<2, 1>(st, 4096, Ws, xa, xb, sink);
    shape<4, 2>(st, 4096, Ws, xa, xb, sink);       // talker qkv: 16.8 MB
    shape<4, 1>(st, 2048, Ws, xa, xb, sink);       // talker o: 8.4 MB
    shape<4, 4>(st, 6144, Ws, xa, xb, sink);       // 25 MB
    return 0;
}
