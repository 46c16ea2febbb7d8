// Development microbench (round 6): the four linears of one talker layer at 32 rows (o_proj -> gate/up -> down -> next qkv; the 9..32-row
// full-K GEMM structure of kernels_lm.hip::k_gemm_fullk: 8 waves split K, every lane requests its weight fragments up front, partial sums meet
// in LDS) as (a) four launches in a hipGraph — the engine's form — and (b) ONE persistent launch of 256 resident blocks with PLAIN-DATA
// hand-offs: a producer stores its rows, releases (agent scope) and sets its flag; a consumer polls the 256 flags (first poll held back),
// acquires and reads the rows through L2.  In (b) the NEXT stage's weight fragments are requested before the hand-off, so they stream while
// the block finishes, publishes and waits.  Simplified arithmetic (no norm / SiLU); outputs of (a) and (b) are compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/layer32_probe tools/layer32_probe.hip && tools/bin/layer32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint16_t bf16_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t as_bf8(uint4 v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ uint4 ldg_nt(const uint4* p) { const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ bf16_t f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
__device__ __forceinline__ size_t frag_off(int r, int k, int K) {
    return ((size_t)(r >> 4) * (K >> 5) + (k >> 5)) * 512 + (size_t)((((k >> 3) & 3) * 16 + (r & 15)) * 8 + (k & 7));
}
// one stage of a block: NB column tiles of 16 starting at tile ct0, rows r0 .. r0 + 16 MT - 1; weights fragment-major [tile][K/32][64 x 16 B]
struct Stage { const bf16_t* W; const bf16_t* xf; bf16_t* yf; bf16_t* y; int N, K, ct0, r0; float scale; };
template <int KS, int NB> struct WRegs { uint4 wv[NB][KS]; };
template <int KS, int NB>
__device__ __forceinline__ void load_w(WRegs<KS, NB>& w, const Stage& s, int wave, int lane) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint4* w0 = reinterpret_cast<const uint4*>(s.W) + (size_t)(s.ct0 + nb) * (s.K >> 5) * 64 + (size_t)wave * KS * 64 + lane;
#pragma unroll
        for (int k = 0; k < KS; ++k) w.wv[nb][k] = ldg_nt(w0 + k * 64);
    }
}
template <int MT, int KS, int NB>
__device__ __forceinline__ void mfma_part(const WRegs<KS, NB>& w, const Stage& s, f32x4_t* red, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    uint4 xa[MT][KS];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const uint4* xr = reinterpret_cast<const uint4*>(s.xf) + (size_t)((s.r0 >> 4) + m) * (s.K >> 5) * 64 + (size_t)wave * KS * 64 + lane;
#pragma unroll
        for (int k = 0; k < KS; ++k) xa[m][k] = xr[k * 64];
    }
    f32x4_t acc[NB][MT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[nb][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(xa[m][k]), as_bf8(w.wv[nb][k]), acc[nb][m], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) red[((wave * NB + nb) * MT + m) * 64 + lane] = acc[nb][m];
}
template <int MT, int NB>
__device__ __forceinline__ void epilogue_part(const Stage& s, const f32x4_t* red, int tid) {
    const int lane = tid & 63;
    __syncthreads();
    if (tid < MT * 64) {
        const int m = tid >> 6, fr = lane & 15;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4_t v = red[((0 * NB + nb) * MT + m) * 64 + lane];
#pragma unroll
            for (int wv = 1; wv < 8; ++wv) v += red[((wv * NB + nb) * MT + m) * 64 + lane];
            const int n = (s.ct0 + nb) * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = s.r0 + m * 16 + (lane >> 4) * 4 + r;
                const bf16_t o = f2bf(v[r] * s.scale);
                if (s.y) s.y[(size_t)b * s.N + n] = o;
                if (s.yf) s.yf[frag_off(b, n, s.N)] = o;
            }
        }
    }
    __syncthreads();      // red may be reused
}
struct Layer { const bf16_t *Wo, *Wc, *Wd, *Wq; const bf16_t* attn; bf16_t *x1, *h, *h2, *x2, *qkv; };
__device__ __forceinline__ Stage st_o(const Layer& L, int b) { return Stage{L.Wo, L.attn, L.x1, nullptr, 2048, 2048, b >> 1, (b & 1) * 16, 0.02f}; }
__device__ __forceinline__ Stage st_c(const Layer& L, int b) {      // 768 tiles of the N = 12288 gate | up matrix: three per block; the second half goes to a dummy
    const int ct = 3 * b;
    return Stage{L.Wc, L.x1, ct < 384 ? L.h : L.h2, nullptr, 6144, 2048, ct < 384 ? ct : ct - 384, 0, 0.02f};
}
__device__ __forceinline__ Stage st_d(const Layer& L, int b) { return Stage{L.Wd, L.h, L.x2, nullptr, 2048, 6144, b >> 1, (b & 1) * 16, 0.01f}; }
__device__ __forceinline__ Stage st_q(const Layer& L, int b) { return Stage{L.Wq, L.x2, nullptr, L.qkv, 4096, 2048, b, 0, 0.02f}; }
// (a) one kernel per stage
template <int WHICH>
__global__ __launch_bounds__(512) void k_stage(Layer L) {
    __shared__ f32x4_t red[8 * 3 * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    if (WHICH == 0) { Stage s = st_o(L, b); WRegs<8, 1> w; load_w(w, s, wave, lane); mfma_part<1, 8, 1>(w, s, red, tid); epilogue_part<1, 1>(s, red, tid); }
    if (WHICH == 1) {
        Stage s = st_c(L, b);
        // (three tiles that straddle the 384 boundary do not occur: 384 = 3 * 128)
        WRegs<8, 3> w; load_w(w, s, wave, lane); mfma_part<2, 8, 3>(w, s, red, tid); epilogue_part<2, 3>(s, red, tid);
    }
    if (WHICH == 2) { Stage s = st_d(L, b); WRegs<24, 1> w; load_w(w, s, wave, lane); mfma_part<1, 24, 1>(w, s, red, tid); epilogue_part<1, 1>(s, red, tid); }
    if (WHICH == 3) { Stage s = st_q(L, b); WRegs<8, 1> w; load_w(w, s, wave, lane); mfma_part<2, 8, 1>(w, s, red, tid); epilogue_part<2, 1>(s, red, tid); }
}
// (b) hand-off: publish this block's flag for `tag`, then wait until all 256 flags carry it
__device__ __forceinline__ void publish(unsigned* flags, unsigned tag, int tid, int b) {
    // (every wave's stores are complete at the barrier in front of this call: epilogue_part ends with __syncthreads)
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flags + b, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void wait_all(const unsigned* flags, unsigned tag, int tid, int delay) {
    if (tid < 256) {
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(2);
        unsigned spins = 0;
        while (__hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag && ++spins < 2000000u) {}
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
template <int PREFETCH>
__global__ __launch_bounds__(512) void k_layer(Layer L, unsigned* flags, unsigned tag0, int delay) {
    __shared__ f32x4_t red[8 * 3 * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const Stage so = st_o(L, b), sc = st_c(L, b), sd = st_d(L, b), sq = st_q(L, b);
    WRegs<8, 1> wo; WRegs<8, 3> wc; WRegs<24, 1> wd; WRegs<8, 1> wq;
    load_w(wo, so, wave, lane);
    mfma_part<1, 8, 1>(wo, so, red, tid);
    if (PREFETCH) load_w(wc, sc, wave, lane);
    epilogue_part<1, 1>(so, red, tid);
    publish(flags, tag0 + 1, tid, b);
    if (!PREFETCH) load_w(wc, sc, wave, lane);
    wait_all(flags, tag0 + 1, tid, delay);
    mfma_part<2, 8, 3>(wc, sc, red, tid);
    if (PREFETCH) load_w(wd, sd, wave, lane);
    epilogue_part<2, 3>(sc, red, tid);
    publish(flags + 256, tag0 + 2, tid, b);
    if (!PREFETCH) load_w(wd, sd, wave, lane);
    wait_all(flags + 256, tag0 + 2, tid, delay);
    mfma_part<1, 24, 1>(wd, sd, red, tid);
    if (PREFETCH) load_w(wq, sq, wave, lane);
    epilogue_part<1, 1>(sd, red, tid);
    publish(flags + 512, tag0 + 3, tid, b);
    if (!PREFETCH) load_w(wq, sq, wave, lane);
    wait_all(flags + 512, tag0 + 3, tid, delay);
    mfma_part<2, 8, 1>(wq, sq, red, tid);
    epilogue_part<2, 1>(sq, red, tid);
}
static const int LAYERS = 14;
template <typename F>
static float time_graph(hipStream_t st, F body, int reps = 20) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    body();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1000.f / (reps * LAYERS);
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t nWo = (size_t)2048 * 2048, nWc = (size_t)12288 * 2048, nWd = (size_t)2048 * 6144, nWq = (size_t)4096 * 2048, nW = nWo + nWc + nWd + nWq;
    std::vector<bf16_t> h(nW);
    uint32_t r = 12345u;
    for (size_t i = 0; i < nW; ++i) { r = r * 1664525u + 1013904223u; h[i] = (bf16_t)(((r >> 16) & 0x80ffu) | 0x3c00u | ((r >> 9) & 0x0300u)); }
    std::vector<bf16_t*> W(LAYERS);
    for (int l = 0; l < LAYERS; ++l) { CK(hipMalloc(&W[l], nW * 2)); CK(hipMemcpy(W[l], h.data() + (l % 3), (nW - 3) * 2, hipMemcpyHostToDevice)); }   // 14 x 92 MB: HBM-cold per layer
    bf16_t *attn, *x1, *hh, *h2, *x2, *qkv, *qkv_ref;
    CK(hipMalloc(&attn, 32 * 2048 * 2)); CK(hipMemcpy(attn, h.data() + 1000, 32 * 2048 * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&x1, 32 * 2048 * 2)); CK(hipMalloc(&hh, 32 * 6144 * 2)); CK(hipMalloc(&h2, 32 * 6144 * 2)); CK(hipMalloc(&x2, 32 * 2048 * 2));
    CK(hipMalloc(&qkv, 32 * 4096 * 2)); CK(hipMalloc(&qkv_ref, 32 * 4096 * 2));
    unsigned* flags; CK(hipMalloc(&flags, 768 * 4)); CK(hipMemset(flags, 0, 768 * 4));
    auto layer = [&](int l, bf16_t* out) { bf16_t* w = W[l]; return Layer{w, w + nWo, w + nWo + nWc, w + nWo + nWc + nWd, attn, x1, hh, h2, x2, out}; };
    // correctness: one layer both ways
    {
        Layer L = layer(0, qkv_ref);
        hipLaunchKernelGGL(k_stage<0>, dim3(256), dim3(512), 0, st, L); hipLaunchKernelGGL(k_stage<1>, dim3(256), dim3(512), 0, st, L);
        hipLaunchKernelGGL(k_stage<2>, dim3(256), dim3(512), 0, st, L); hipLaunchKernelGGL(k_stage<3>, dim3(256), dim3(512), 0, st, L);
        CK(hipStreamSynchronize(st));
        std::vector<bf16_t> a(32 * 4096), c(32 * 4096);
        CK(hipMemcpy(a.data(), qkv_ref, a.size() * 2, hipMemcpyDeviceToHost));
        for (int pf = 0; pf < 2; ++pf) {
            CK(hipMemsetAsync(x1, 0, 32 * 2048 * 2, st)); CK(hipMemsetAsync(hh, 0, 32 * 6144 * 2, st)); CK(hipMemsetAsync(x2, 0, 32 * 2048 * 2, st));
            Layer L2 = layer(0, qkv);
            if (pf) hipLaunchKernelGGL(k_layer<1>, dim3(256), dim3(512), 0, st, L2, flags, 8u * (1 + pf), 8);
            else hipLaunchKernelGGL(k_layer<0>, dim3(256), dim3(512), 0, st, L2, flags, 8u * (1 + pf), 8);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(c.data(), qkv, c.size() * 2, hipMemcpyDeviceToHost));
            size_t bad = 0, nz = 0;
            for (size_t i = 0; i < a.size(); ++i) { bad += a[i] != c[i]; nz += (a[i] & 0x7fff) != 0; }
            printf("persistent (prefetch %d) vs four launches: %zu of %zu outputs differ (%zu nonzero)\n", pf, bad, a.size(), nz);
        }
    }
    const float chain = time_graph(st, [&]() {
        for (int l = 0; l < LAYERS; ++l) {
            Layer L = layer(l, qkv);
            hipLaunchKernelGGL(k_stage<0>, dim3(256), dim3(512), 0, st, L); hipLaunchKernelGGL(k_stage<1>, dim3(256), dim3(512), 0, st, L);
            hipLaunchKernelGGL(k_stage<2>, dim3(256), dim3(512), 0, st, L); hipLaunchKernelGGL(k_stage<3>, dim3(256), dim3(512), 0, st, L);
        }
    });
    printf("four launches per layer (graph):                 %7.2f us per layer (92.3 MB of weights: %.2f TB/s)\n", chain, 92.3 / chain);
    for (int pf = 0; pf < 2; ++pf)
        for (int delay : {0, 4, 8, 16, 24}) {
            const float t = time_graph(st, [&]() {
                CK(hipMemsetAsync(flags, 0, 768 * 4, st));
                for (int l = 0; l < LAYERS; ++l) {
                    Layer L = layer(l, qkv);
                    if (pf) hipLaunchKernelGGL(k_layer<1>, dim3(256), dim3(512), 0, st, L, flags, 8u * (l + 1), delay);
                    else hipLaunchKernelGGL(k_layer<0>, dim3(256), dim3(512), 0, st, L, flags, 8u * (l + 1), delay);
                }
            });
            printf("one persistent launch per layer, prefetch %d, first poll held back %2d x 128 clk: %7.2f us per layer (%.2f TB/s)\n", pf, delay, t, 92.3 / t);
        }
    return 0;
}
