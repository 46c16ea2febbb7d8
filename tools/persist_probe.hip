// Development microbench: what does ONE dependent GEMV stage cost inside a persistent kernel on MI355X, against the same chain
// as hipGraph-captured launches?  The depth loop of the frame is ~330 dependent stages of 4-12 MB (cache-resident weights):
// 4.6-5.2 us each as separate launches.  Here: 256 blocks x 512 threads stay resident; stage s computes y = W_s x (N = 2048
// outputs, one per wave, K = 1024), publishes y as 8-byte {tag, 2 x bf16} granules (relaxed agent-scope atomic stores = sc1
// write-through, MI355X guide Guideline 16 form R2), every block re-gathers the whole vector (relaxed agent-scope 8-byte loads,
// swept until every tag matches) into LDS; the NEXT stage's weights are requested before the gather so that they are in
// registers when x arrives.  Bounded spins; granule words zeroed before every launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/persist_probe tools/persist_probe.hip && tools/bin/persist_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint16_t bf16_t;
typedef unsigned long long u64;
__device__ __forceinline__ float bflo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bfhi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ bf16_t f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
__device__ __forceinline__ float dot8(uint4 w, uint4 x, float s) {
    s = fmaf(bflo(w.x), bflo(x.x), s); s = fmaf(bfhi(w.x), bfhi(x.x), s); s = fmaf(bflo(w.y), bflo(x.y), s); s = fmaf(bfhi(w.y), bfhi(x.y), s);
    s = fmaf(bflo(w.z), bflo(x.z), s); s = fmaf(bfhi(w.z), bfhi(x.z), s); s = fmaf(bflo(w.w), bflo(x.w), s); s = fmaf(bfhi(w.w), bfhi(x.w), s);
    return s;
}
__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

#ifndef K_
#define K_ 1024
#endif
constexpr int K = K_, N = 2048, NBLK = 256, NTHR = 512, NGRAN = N / 2, KC = K / 512;   // KC uint4 chunks per lane
constexpr int GPT = K / 2 / NTHR;                                                     // granules gathered per thread
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Args {
    const bf16_t* W;      // [n_w][N][K]
    int n_w, stages;
    u64* gran;            // [2][NGRAN]  (two buffers, alternating by stage parity)
    const bf16_t* x0;     // [K]
    bf16_t* y_out;        // [N] of the last stage
    unsigned* err;        // spin timeout flag
    float scale;
};

// x of a stage = the first K entries of the previous stage's y (N >= K), scaled
__global__ __launch_bounds__(NTHR) void k_persist(Args a) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[K];
    __shared__ bf16_t ys[NTHR / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = blockIdx.x * (NTHR / 64) + wave;
    for (int i = tid; i < K; i += NTHR) xs[i] = a.x0[i];
    uint4 w[KC];
    {
        const uint4* wr = reinterpret_cast<const uint4*>(a.W + (size_t)col * K);
#pragma unroll
        for (int j = 0; j < KC; ++j) w[j] = wr[lane + 64 * j];
    }
    __syncthreads();
    for (int s = 0; s < a.stages; ++s) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < KC; ++j) acc = dot8(w[j], reinterpret_cast<const uint4*>(xs)[lane + 64 * j], acc);
        acc = wave_sum(acc) * a.scale;
        if (lane == 0) ys[wave] = f2bf(acc);
        // next stage's weights: in flight during the publish + gather
        if (s + 1 < a.stages) {
            const uint4* wr = reinterpret_cast<const uint4*>(a.W + ((size_t)((s + 1) % a.n_w) * N + col) * K);
#pragma unroll
            for (int j = 0; j < KC; ++j) w[j] = wr[lane + 64 * j];
        }
        __syncthreads();
        u64* g = a.gran + (size_t)(s & 1) * NGRAN;
        const unsigned tag = (unsigned)s + 1u;
        if (tid < NTHR / 128) {      // 4 granules per block: columns (2 tid, 2 tid + 1)
            const unsigned v = (unsigned)ys[2 * tid] | ((unsigned)ys[2 * tid + 1] << 16);
            __hip_atomic_store(g + blockIdx.x * (NTHR / 128) + tid, ((u64)tag << 32) | v, RLX_AGENT);
        }
        if (s + 1 == a.stages) {
            if (lane == 0) a.y_out[col] = ys[wave];
            break;
        }
        // gather the first K / 2 granules (the next x): thread t takes granules t, t + 512, ...; re-read until every tag matches
        unsigned val[GPT];
        bool ok = false;
        for (unsigned spins = 0; !ok; ++spins) {
            ok = true;
#pragma unroll
            for (int q = 0; q < GPT; ++q) {
                const u64 x = __hip_atomic_load(g + tid + q * NTHR, RLX_AGENT);
                ok = ok && (unsigned)(x >> 32) == tag;
                val[q] = (unsigned)x;
            }
            if (spins > 2000000u) { if (tid == 0) atomicExch(a.err, 1u + (unsigned)s); ok = true; }
        }
#pragma unroll
        for (int q = 0; q < GPT; ++q) reinterpret_cast<unsigned*>(xs)[tid + q * NTHR] = val[q];
        __syncthreads();
    }
}

// the same chain as separate launches (hipGraph): y = W_s x -> global; x read plainly
__global__ __launch_bounds__(NTHR) void k_stage(const bf16_t* W, const bf16_t* x, bf16_t* y, float scale) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = blockIdx.x * (NTHR / 64) + wave;
    const uint4* wr = reinterpret_cast<const uint4*>(W + (size_t)col * K);
    const uint4* xr = reinterpret_cast<const uint4*>(x);
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < KC; ++j) acc = dot8(wr[lane + 64 * j], xr[lane + 64 * j], acc);
    acc = wave_sum(acc) * scale;
    if (lane == 0) y[col] = f2bf(acc);
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 300;
    const int n_w = argc > 2 ? atoi(argv[2]) : 40;       // 40 x 4 MB = 160 MB of weights cycling: Infinity-Cache resident, not L2
    hipStream_t st; CK(hipStreamCreate(&st));
    bf16_t *W, *x0, *y, *ya, *yb; u64* gran; unsigned* err;
    CK(hipMalloc(&W, (size_t)n_w * N * K * 2)); CK(hipMalloc(&x0, K * 2)); CK(hipMalloc(&y, N * 2)); CK(hipMalloc(&ya, N * 2)); CK(hipMalloc(&yb, N * 2));
    CK(hipMalloc(&gran, 2 * NGRAN * 8)); CK(hipMalloc(&err, 4));
    {
        std::vector<bf16_t> h((size_t)n_w * N * K);
        uint32_t r = 12345u;
        for (size_t i = 0; i < h.size(); ++i) { r = r * 1664525u + 1013904223u; const float f = ((int)((r >> 8) & 0xffff) - 32768) / 32768.0f * 0.06f; uint32_t u; memcpy(&u, &f, 4); h[i] = (bf16_t)(u >> 16); }
        CK(hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        std::vector<bf16_t> hx(K);
        for (int i = 0; i < K; ++i) { const float f = 0.5f + 0.001f * i; uint32_t u; memcpy(&u, &f, 4); hx[i] = (bf16_t)(u >> 16); }
        CK(hipMemcpy(x0, hx.data(), K * 2, hipMemcpyHostToDevice));
    }
    const float scale = K == 1024 ? 0.85f : 0.6f;
    // ---- launches baseline: hipGraph chain
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < stages; ++s) {
        const bf16_t* xin = s == 0 ? x0 : ((s & 1) ? ya : yb);
        bf16_t* yout = (s & 1) ? yb : ya;
        hipLaunchKernelGGL(k_stage, dim3(NBLK), dim3(NTHR), 0, st, W + (size_t)(s % n_w) * N * K, xin, yout, scale);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launch chain : %d stages, %.3f us per stage\n", stages, ms * 1000.f / (10 * stages));
    std::vector<bf16_t> ref(N);
    CK(hipMemcpy(ref.data(), ((stages - 1) & 1) ? yb : ya, N * 2, hipMemcpyDeviceToHost));
    // ---- persistent
    Args a{W, n_w, stages, gran, x0, y, err, scale};
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(gran, 0, 2 * NGRAN * 8, st)); CK(hipMemsetAsync(err, 0, 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_persist, dim3(NBLK), dim3(NTHR), 0, st, a);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<bf16_t> got(N);
    CK(hipMemcpy(got.data(), y, N * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; ++i) bad += got[i] != ref[i];
    printf("persistent   : %d stages, %.3f us per stage (best of 5), timeout flag %u, mismatches vs the launch chain %d / %d\n",
           stages, best * 1000.f / stages, herr, bad, N);
    return 0;
}
