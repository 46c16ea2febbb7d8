// Development microbench: where does the time of the 9..32-row full-K GEMM (k_gemm_fullk) go?
// The kernel body below has the same load / MFMA / reduce / epilogue structure as vox_serve_amd/csrc/kernels_lm.hip::k_gemm_fullk
// (copy prologue, plain store epilogue, fragment-major operands), with switches that strip one ingredient at a time.
// Each variant is timed as a hipGraph chain of CHAIN dependent launches over rotating (HBM-cold) weight buffers.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/fullk_probe tools/fullk_probe.hip && tools/bin/fullk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint16_t bf16_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t as_bf8(uint4 v) { return __builtin_bit_cast(bf16x8_t, v); }
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldg_nt(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bf16_t f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ size_t frag_off(int r, int k, int K) {
    return ((size_t)(r >> 4) * (K >> 5) + (k >> 5)) * 512 + (size_t)((((k >> 3) & 3) * 16 + (r & 15)) * 8 + (k & 7));
}
struct Args {
    const bf16_t *W, *x, *res;
    bf16_t *y, *yf;
    int B, N, K, row_tiles;
};
enum { F_W = 1, F_A = 2, F_RES = 4, F_YF = 8, F_NT = 16, F_RESPRE = 32, F_MFMA = 64, F_Y = 128 };
// WAVES waves split K; KSTEPS 32-wide steps per wave; MT 16-row tiles; NB column tiles of 16 per block
template <int WAVES, int MT, int KSTEPS, int NB, int FL, int ID = 0>
__global__ __launch_bounds__(WAVES * 64) void k_probe(Args a) {
    if (a.B == -1 - ID) a.y[ID] = ID;      // ID: distinct copies of the same code (instruction-cache experiments)
    __shared__ f32x4_t red[WAVES][NB][MT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15;
    int ctile = blockIdx.x, r0 = 0;
    if (a.row_tiles > 1) {
        ctile = (blockIdx.x / (8 * a.row_tiles)) * 8 + (blockIdx.x & 7);
        r0 = ((blockIdx.x >> 3) % a.row_tiles) * (16 * MT);
    }
    const int bt = a.B - r0;
    const size_t fbase = (size_t)wave * KSTEPS * 64 + lane;
    uint4 wv[NB][KSTEPS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint4* w0 = reinterpret_cast<const uint4*>(a.W) + (size_t)(ctile * NB + nb) * (a.K >> 5) * 64 + fbase;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            if (FL & F_W) wv[nb][s] = (FL & F_NT) ? ldg_nt(w0 + s * 64) : w0[s * 64];
            else wv[nb][s] = make_uint4(0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u + s, 0x3f803f80u);
        }
    }
    float rpre[4] = {0.f, 0.f, 0.f, 0.f};
    if ((FL & F_RES) && (FL & F_RESPRE) && tid < MT * 64) {
        const int m = tid >> 6;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = m * 16 + (lane >> 4) * 4 + r;
            if (b < bt) rpre[r] = bf2f(a.res[(size_t)(r0 + b) * a.N + ctile * 16 + fr]);
        }
    }
    uint4 xa[MT][KSTEPS];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const uint4* xr = reinterpret_cast<const uint4*>(a.x) + (size_t)((r0 >> 4) + m) * (a.K >> 5) * 64 + fbase;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            if (FL & F_A) xa[m][s] = xr[s * 64];
            else xa[m][s] = make_uint4(0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u + m);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t acc[NB][MT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[nb][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (FL & F_MFMA) acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(xa[m][s]), as_bf8(wv[nb][s]), acc[nb][m], 0, 0, 0);
                else acc[nb][m][0] += __uint_as_float(xa[m][s].x ^ wv[nb][s].y ^ xa[m][s].z ^ wv[nb][s].w ^ xa[m][s].y ^ wv[nb][s].x ^ xa[m][s].w ^ wv[nb][s].z);
            }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) red[wave][nb][m][lane] = acc[nb][m];
    __syncthreads();
    if (tid >= MT * 64) return;
    const int m = tid >> 6;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        f32x4_t v = red[0][nb][m][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += red[w][nb][m][lane];
        const int n = (ctile * NB + nb) * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = m * 16 + (lane >> 4) * 4 + r;
            if (b >= bt) continue;
            const size_t oi = (size_t)(r0 + b) * a.N + n;
            bf16_t o = f2bf(v[r]);
            if (FL & F_RES) o = f2bf(((FL & F_RESPRE) ? rpre[r] : bf2f(a.res[oi])) + bf2f(o));
            if (FL & F_Y) a.y[oi] = o;
            if (FL & F_YF) a.yf[frag_off(r0 + b, n, a.N)] = o;
        }
    }
}
__global__ void k_empty(Args a) { if (a.B < 0) a.y[0] = 0; }

struct Bufs { std::vector<bf16_t*> W; bf16_t *x, *res, *y, *yf; };
static const int CHAIN = 56;

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
    const int reps = 20;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1000.f / (reps * CHAIN);
}

template <int WAVES, int MT, int KSTEPS, int NB, int FL>
static void run(const char* name, hipStream_t st, Bufs& b, int B, int N, int K, int row_tiles) {
    if (K != WAVES * KSTEPS * 32) { printf("bad K for %s\n", name); return; }
    const int grid = N / (16 * NB) * row_tiles;
    float us = time_chain(st, [&](int i) {
        const size_t slots = ((size_t)2 * 6144 * 2048) / ((size_t)N * K);      // distinct sub-buffers: the chain touches > 256 MB (MALL) of weights
        Args a{b.W[i % b.W.size()] + ((i / b.W.size()) % slots) * (size_t)N * K, b.x, b.res, b.y, b.yf, B, N, K, row_tiles};
        hipLaunchKernelGGL((k_probe<WAVES, MT, KSTEPS, NB, FL>), dim3(grid), dim3(WAVES * 64), 0, st, a);
    });
    printf("  %-46s grid %4d x %4d  %7.2f us  %5.2f TB/s(W)\n", name, grid, WAVES * 64, us, (double)N * K * 2 / us / 1e6);
}

int main(int argc, char** argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t maxW = (size_t)2 * 6144 * 2048;   // gate+up
    Bufs b;
    const int nbuf = 28;
    for (int i = 0; i < nbuf; ++i) { bf16_t* p; CK(hipMalloc(&p, maxW * 2)); CK(hipMemset(p, 0x3c, maxW * 2)); b.W.push_back(p); }
    CK(hipMalloc(&b.x, 128 * 8192 * 2)); CK(hipMemset(b.x, 0x3c, 128 * 8192 * 2));
    CK(hipMalloc(&b.res, 128 * 16384 * 2)); CK(hipMemset(b.res, 0x3c, 128 * 16384 * 2));
    CK(hipMalloc(&b.y, 128 * 16384 * 2)); CK(hipMalloc(&b.yf, 128 * 16384 * 2));
    if (argc > 1) {     // random bf16 data instead of a constant fill
        std::vector<bf16_t> h(maxW);
        uint32_t r = 12345u;
        for (size_t i = 0; i < maxW; ++i) { r = r * 1664525u + 1013904223u; h[i] = (bf16_t)(((r >> 16) & 0x80ffu) | 0x3c00u | ((r >> 9) & 0x0300u)); }
        for (int i = 0; i < nbuf; ++i) CK(hipMemcpy(b.W[i], h.data(), maxW * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.x, h.data(), 128 * 8192 * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.res, h.data() + 77, 128 * 16384 * 2, hipMemcpyHostToDevice));
        printf("random data\n");
    }
    {
        float us = time_chain(st, [&](int i) { Args a{b.W[0], b.x, b.res, b.y, b.yf, 32, 2048, 2048, 1}; hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, a); });
        printf("empty kernel 256 x 512: %.2f us per launch\n", us);
    }
    constexpr int ALL = F_W | F_A | F_RES | F_YF | F_NT | F_MFMA | F_Y;
    printf("o_proj  B=32 N=2048 K=2048  (engine: <1,8,COPY,STORE,ROWSPLIT> 12.6 us)\n");
    run<8, 1, 8, 1, ALL>("as engine (rowsplit 2x16 rows)", st, b, 32, 2048, 2048, 2);
    run<8, 1, 8, 1, ALL | F_RESPRE>("+ residual prefetched", st, b, 32, 2048, 2048, 2);
    run<8, 1, 8, 1, ALL & ~F_RES>("- residual", st, b, 32, 2048, 2048, 2);
    run<8, 1, 8, 1, ALL & ~F_YF>("- y_frag store", st, b, 32, 2048, 2048, 2);
    run<8, 1, 8, 1, ALL & ~F_A>("- A loads", st, b, 32, 2048, 2048, 2);
    run<8, 1, 8, 1, ALL & ~F_W>("- W loads", st, b, 32, 2048, 2048, 2);
    run<8, 1, 8, 1, ALL & ~F_NT>("- nt (plain W loads)", st, b, 32, 2048, 2048, 2);
    run<8, 1, 8, 1, ALL & ~F_MFMA>("- mfma", st, b, 32, 2048, 2048, 2);
    run<8, 1, 8, 1, F_MFMA | F_Y>("no loads at all", st, b, 32, 2048, 2048, 2);
    run<8, 2, 8, 1, ALL>("one 32-row block per tile (grid 128)", st, b, 32, 2048, 2048, 1);
    run<4, 1, 16, 1, ALL | F_RESPRE>("4 waves x 16 ksteps, rowsplit, res pre", st, b, 32, 2048, 2048, 2);
    run<4, 2, 16, 1, ALL | F_RESPRE>("4 waves x 16 ksteps, 32 rows, res pre", st, b, 32, 2048, 2048, 1);
    run<16, 1, 4, 1, ALL | F_RESPRE>("16 waves x 4 ksteps, rowsplit, res pre", st, b, 32, 2048, 2048, 2);
    run<16, 2, 4, 1, ALL | F_RESPRE>("16 waves x 4 ksteps, 32 rows, res pre", st, b, 32, 2048, 2048, 1);
    printf("down    B=32 N=2048 K=6144  (engine: <1,24,COPY,STORE,ROWSPLIT> 14.7 us)\n");
    run<8, 1, 24, 1, ALL>("as engine", st, b, 32, 2048, 6144, 2);
    run<8, 1, 24, 1, ALL | F_RESPRE>("+ residual prefetched", st, b, 32, 2048, 6144, 2);
    run<8, 1, 24, 1, ALL & ~F_A>("- A loads", st, b, 32, 2048, 6144, 2);
    run<8, 1, 24, 1, ALL & ~F_W>("- W loads", st, b, 32, 2048, 6144, 2);
    run<16, 1, 12, 1, ALL | F_RESPRE>("16 waves x 12 ksteps, rowsplit, res pre", st, b, 32, 2048, 6144, 2);
    run<16, 2, 12, 1, ALL | F_RESPRE>("16 waves x 12 ksteps, 32 rows, res pre", st, b, 32, 2048, 6144, 1);
    run<8, 2, 24, 1, ALL | F_RESPRE>("8 waves, 32 rows (grid 128), res pre", st, b, 32, 2048, 6144, 1);
    printf("qkv     B=32 N=4096 K=2048 (engine adds the norm prologue: <2,8,NORM,STORE> 16.8 us)\n");
    run<8, 2, 8, 1, ALL & ~F_RES>("copy prologue, 32 rows, grid 256", st, b, 32, 4096, 2048, 1);
    run<8, 2, 8, 1, ALL & ~F_RES & ~F_A>("- A loads", st, b, 32, 4096, 2048, 1);
    run<8, 2, 8, 1, ALL & ~F_RES & ~F_W>("- W loads", st, b, 32, 4096, 2048, 1);
    run<8, 1, 8, 1, ALL & ~F_RES>("rowsplit (grid 512)", st, b, 32, 4096, 2048, 2);
    run<16, 2, 4, 1, ALL & ~F_RES>("16 waves x 4 ksteps, 32 rows", st, b, 32, 4096, 2048, 1);
    run<16, 1, 4, 1, ALL & ~F_RES>("16 waves x 4 ksteps, rowsplit", st, b, 32, 4096, 2048, 2);
    printf("gate/up B=32 N=2x6144 K=2048 as one N=12288 GEMM (engine: <2,8,NORM,SILU_MUL> 25.2 us, 247 VGPR)\n");
    run<8, 2, 8, 2, ALL & ~F_RES>("copy prologue, 32 rows, NB=2, grid 384", st, b, 32, 12288, 2048, 1);
    run<8, 2, 8, 1, ALL & ~F_RES>("NB=1, grid 768", st, b, 32, 12288, 2048, 1);
    run<8, 1, 8, 2, ALL & ~F_RES>("rowsplit NB=2 (grid 768)", st, b, 32, 12288, 2048, 2);
    run<8, 1, 8, 1, ALL & ~F_RES>("rowsplit NB=1 (grid 1536)", st, b, 32, 12288, 2048, 2);
    run<16, 2, 4, 2, ALL & ~F_RES>("16 waves x 4 ksteps NB=2 (grid 384)", st, b, 32, 12288, 2048, 1);
    run<16, 2, 4, 1, ALL & ~F_RES>("16 waves x 4 ksteps NB=1 (grid 768)", st, b, 32, 12288, 2048, 1);
    run<8, 2, 8, 2, ALL & ~F_RES & ~F_A>("NB=2 - A loads", st, b, 32, 12288, 2048, 1);
    run<8, 2, 8, 2, ALL & ~F_RES & ~F_W>("NB=2 - W loads", st, b, 32, 12288, 2048, 1);
    // --- instruction-cache experiments: the engine's frame alternates ~12 different kernels per layer, this chain so far ran one
    printf("o_proj shape, chain cycling through D distinct copies of the same kernel code\n");
    {
        auto cyc = [&](int D) {
            const int N = 2048, K = 2048;
            float us = time_chain(st, [&](int i) {
                const size_t slots = ((size_t)2 * 6144 * 2048) / ((size_t)N * K);
                Args a{b.W[i % b.W.size()] + ((i / b.W.size()) % slots) * (size_t)N * K, b.x, b.res, b.y, b.yf, 32, N, K, 2};
                constexpr int AL = F_W | F_A | F_RES | F_YF | F_NT | F_MFMA | F_Y;
                switch (i % D) {
#define C_(J) case J: hipLaunchKernelGGL((k_probe<8, 1, 8, 1, AL, 100 + J>), dim3(256), dim3(512), 0, st, a); break;
                    C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15)
                    C_(16) C_(17) C_(18) C_(19) C_(20) C_(21) C_(22) C_(23) C_(24) C_(25) C_(26) C_(27) C_(28) C_(29) C_(30) C_(31)
#undef C_
                }
            });
            printf("  %2d copies: %7.2f us per launch\n", D, us);
        };
        cyc(1); cyc(2); cyc(4); cyc(8); cyc(16); cyc(32);
    }
    printf("one talker layer's four GEMMs in sequence (o, gate/up, down, qkv), 14 layers per graph\n");
    {
        float us = time_chain(st, [&](int i) {
            const int which = i & 3;
            constexpr int AL = F_W | F_A | F_YF | F_NT | F_MFMA | F_Y;
            bf16_t* W = b.W[(i >> 2) % b.W.size()];
            if (which == 0) { Args a{W, b.x, b.res, b.y, b.yf, 32, 2048, 2048, 2}; hipLaunchKernelGGL((k_probe<8, 1, 8, 1, AL | F_RES>), dim3(256), dim3(512), 0, st, a); }
            if (which == 1) { Args a{W, b.x, b.res, b.y, b.yf, 32, 12288, 2048, 1}; hipLaunchKernelGGL((k_probe<8, 2, 8, 2, AL>), dim3(384), dim3(512), 0, st, a); }
            if (which == 2) { Args a{W, b.x, b.res, b.y, b.yf, 32, 2048, 6144, 2}; hipLaunchKernelGGL((k_probe<8, 1, 24, 1, AL | F_RES>), dim3(256), dim3(512), 0, st, a); }
            if (which == 3) { Args a{W, b.x, b.res, b.y, b.yf, 32, 4096, 2048, 1}; hipLaunchKernelGGL((k_probe<8, 2, 8, 1, AL>), dim3(256), dim3(512), 0, st, a); }
        });
        printf("  %7.2f us per launch = %7.2f us per layer (single-kernel chains above: o + gate/up + down + qkv)\n", us, us * 4);
    }
    return 0;
}
