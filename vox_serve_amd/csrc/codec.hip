// Qwen3-TTS 12 Hz codec decoder (token -> waveform), streaming, for gfx950.
// Replaces Qwen3TTSTokenizerV2Decoder.forward_chunk (vox_serve/tokenizer/qwen3_codec.py:1541-1666) and the
// per-request DecoderCache cat / copy-in / copy-out of CudaGraphWorker.run_detokenize
// (worker/cuda_graph_worker.py:1217-1241: ~57 MiB per request per chunk) with in-place per-slot state
// (conv tails + a 72-slot bf16 KV ring: ~2.6 MiB per request).
//
// Layout: activations fp32, time-major [rows = (request, t)][channels] so that
//   * every conv / transposed conv / linear is ONE implicit GEMM  out[t, n] = sum_tap sum_c A[t - off_tap, c] * W[tap][n][c]
//     on the matrix cores.  The waveform parity bar of 1e-4 RMS is not reachable with bf16-rounded activations, so the
//     fp32 activation is split EXACTLY into three bf16 terms when it is staged (x = h + m + l, 24 mantissa bits) and each
//     term meets the (already bf16) weight in v_mfma_f32_16x16x32_bf16: products exact, fp32 accumulation — the result
//     of an fp32-input MFMA at 3/16 of its issue cost (the fp32 16x16x4 form made these kernels MFMA-bound per CU);
//   * a transposed conv with stride r writes r*Cout contiguous values per input row = r output rows: no scatter;
//   * LayerNorm / RMSNorm / depthwise conv run over the contiguous channel axis.
// Weights stay bf16 in HBM ([tap][N][Cin], packed once on the host side) and are widened when staged to LDS.
#include "vox_internal.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ cbf16x8 as_cbf8(uint4 v) {
    union { uint4 u; cbf16x8 b; } c;
    c.u = v;
    return c.b;
}
// exact three-way split of 8 fp32 values into bf16 planes: x = h + m + l (each subtraction is exact in fp32)
__device__ __forceinline__ void split3(const float4 lo4, const float4 hi4, uint4& h, uint4& m, uint4& l) {
    const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    unsigned short hb[8], mb[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hb[i] = f2bf(x[i]);
        const float r1 = x[i] - bf2f(hb[i]);
        mb[i] = f2bf(r1);
        const float r2 = r1 - bf2f(mb[i]);
        lb[i] = f2bf(r2);
    }
    h = make_uint4(hb[0] | (unsigned)hb[1] << 16, hb[2] | (unsigned)hb[3] << 16, hb[4] | (unsigned)hb[5] << 16, hb[6] | (unsigned)hb[7] << 16);
    m = make_uint4(mb[0] | (unsigned)mb[1] << 16, mb[2] | (unsigned)mb[3] << 16, mb[4] | (unsigned)mb[5] << 16, mb[6] | (unsigned)mb[7] << 16);
    l = make_uint4(lb[0] | (unsigned)lb[1] << 16, lb[2] | (unsigned)lb[3] << 16, lb[4] | (unsigned)lb[5] << 16, lb[6] | (unsigned)lb[7] << 16);
}

// ================================================================================================
// implicit-GEMM convolution on the matrix cores (fp32 activations split 3-way into bf16).  Block 256 threads = 2x2 waves.
// ================================================================================================
#define CG_BM 64
#define CG_BN 64
#define CG_BK 32            // smallest K chunk (Cin granularity); 64-wide chunks are used where Cin allows
#define CG_MAXTAPS 24          // (HiFT: 11-tap dilated convs with fp32 weights carried as two bf16 planes)

struct ConvGemmArgs {
    const float* x;        // [n*L, Cin] current-chunk input rows
    const float* state;    // [slots, P, Cin] history rows (NULL when no tap looks back)
    const int* slots;      // [n] state slot of each request in this batch
    const bf16_t* w;       // [taps][N][Cin]
    const float* bias;     // [bias_mod] or NULL
    const float* res;      // [M, N] residual or NULL
    const float* scale;    // [N] multiplier applied before the residual add, or NULL
    const float* rscale;   // [M] per-row multiplier applied before the residual add, or NULL (SNAC NoiseBlock: x + noise[t] * Wx)
    int planes;            // 3: exact products (fp32 activation = three bf16 terms); 1: activations rounded to bf16 (the reference's
                           // own serving precision: its decoder runs in bf16, qwen3_tts.py:1061-1064) at a third of the MFMA issue
    float* out;            // [M, N]
    int M, N, Cin, L, P, n_taps, bias_mod, gelu;
    int off[CG_MAXTAPS];   // row look-back of each tap (negative: look-ahead)
    // fused SnakeBeta of the NEXT layer's input: out2[m][n] = v + inv_beta[c] * sin(v * alpha[c])^2, c = n % sn_mod, beside
    // (out != NULL) or instead of (out == NULL) the plain output — the arithmetic of k_snake, one elementwise pass less
    float* out2;
    const float *sn_alpha, *sn_invb;
    int sn_mod;
    // LayerNorm of the input rows fused into the A-operand staging (few-row kernel, one tap): x <- (x - mean) * rstd * ln_w + ln_b
    const float *ln_w, *ln_b;
    float ln_eps;
    // "S2" rows: an activation row of Cin values stored as its two leading bf16 terms, [Cin x bf16 h][Cin x bf16 m] (x ~ h + m, 16
    // significand bits; the same 4 * Cin bytes as the fp32 row, so buffers, history state and k_state_update are format-agnostic).
    // The producing GEMM's epilogue splits each value ONCE (out2_s2); the consuming GEMM (x_s2, planes = 2) stages plain bf16 — the
    // per-tap three-way split of fp32 rows made the many-row decoder convs VALU-bound (MFMA 16 % busy; round-3 planes probe).
    int x_s2, out2_s2;
    int dev;               // development builds (-DVOX_DEV_KNOBS, VOX_CODEC_DEV): 1 skip the K loop, 2 no out2 stores, 4 no out stores, 8 no residual loads
};
#ifdef VOX_DEV_KNOBS
#define CG_DEV(a, bit) ((a).dev & (bit))
#else
#define CG_DEV(a, bit) 0
#endif
// the two leading bf16 terms of v
__device__ __forceinline__ void split2(float v, bf16_t& h, bf16_t& m) {
    h = f2bf(v);
    m = f2bf(v - bf2f(h));
}
// sin for the fused epilogues: explicit reduction to revolutions + the hardware v_sin_f32 (absolute error ~1e-6 on [-1, 1]:
// far inside the 1e-4 waveform bar, and sin^2 enters scaled by 1/beta ~ 1).  The libm sinf (argument-reduction table,
// private array) cannot be inlined into a GEMM epilogue without spilling the accumulators to scratch.
__device__ __forceinline__ float snake_f(float v, float alpha, float invb) {
    const float r = (v * alpha) * 0.15915494309189535f;
    const float s_ = __builtin_amdgcn_sinf(r - floorf(r));
    return v + invb * (s_ * s_);
}

// Epilogue of the tiled conv GEMMs (2 x 2 waves, WM x WN 16x16 tiles per wave)
template <int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvGemmArgs& a, f32x4 (&acc)[WM][WN], int m0, int n0, int wm, int wn, int lane) {
    // epilogue.  D layout of a 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg.  The per-column constants are fetched once and
    // the residual values of a whole row fragment are requested together, ahead of the arithmetic: an epilogue that interleaves loads,
    // stores and integer divisions element by element serialises on memory latency (the compiler must keep every load behind the
    // previous store) — the 1-tap conv2 of a residual unit took longer than its 7-tap conv1 (485 vs 365 us at 614 k rows).
    const int l15 = lane & 15, rg = (lane >> 4) * 4;
    float cb[WN], cs[WN], ca[WN], ci[WN];
    int cn[WN], cj[WN], cc[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + wn + j * 16 + l15, nn = n < a.N ? n : a.N - 1;
        cn[j] = n;
        cb[j] = a.bias ? a.bias[nn % a.bias_mod] : 0.0f;
        cs[j] = a.scale ? a.scale[nn] : 1.0f;
        ca[j] = ci[j] = 0.0f; cj[j] = cc[j] = 0;
        if (a.out2) {
            cj[j] = nn / a.sn_mod; cc[j] = nn - cj[j] * a.sn_mod;      // (a transposed conv's N = r * sn_mod outputs: r consumer rows)
            ca[j] = a.sn_alpha[cc[j]]; ci[j] = a.sn_invb[cc[j]];
        }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int mb = m0 + wm + i * 16 + rg;
        float rs[WN][4], rsc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rsc[r] = (a.rscale && mb + r < a.M) ? a.rscale[mb + r] : 1.0f;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rs[j][r] = (a.res && !CG_DEV(a, 8) && mb + r < a.M && cn[j] < a.N) ? a.res[(size_t)(mb + r) * a.N + cn[j]] : 0.0f;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + r;
                const bool ok = m < a.M && cn[j] < a.N;
                float v = acc[i][j][r] + cb[j];
                if (a.gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                if (a.rscale) v = rsc[r] * v;
                if (a.res) v = rs[j][r] + cs[j] * v;
                else if (a.scale) v = cs[j] * v;
                const size_t o = (size_t)m * a.N + cn[j];
                if (a.out && ok && !CG_DEV(a, 4)) a.out[o] = v;
                if (a.out2 && !CG_DEV(a, 2)) {
                    const float sv2 = snake_f(v, ca[j], ci[j]);
                    if (a.out2_s2) {
                        // S2 row of the consumer: [sn_mod x h][sn_mod x m]; a lane pair (columns c, c + 1) stores one 32-bit word per
                        // term (the even lane the h pair, the odd lane the m pair) instead of two 16-bit stores each
                        bf16_t hb, mbf;
                        split2(sv2, hb, mbf);
                        const unsigned hp = __shfl_xor((unsigned)hb, 1, 64), mp = __shfl_xor((unsigned)mbf, 1, 64);
                        bf16_t* r16 = reinterpret_cast<bf16_t*>(a.out2 + (size_t)m * a.N) + (size_t)cj[j] * 2 * a.sn_mod;
                        if (ok) {
                            if (!(l15 & 1)) *reinterpret_cast<unsigned*>(r16 + cc[j]) = (unsigned)hb | (hp << 16);
                            else *reinterpret_cast<unsigned*>(r16 + a.sn_mod + cc[j] - 1) = mp | ((unsigned)mbf << 16);
                        }
                    } else if (ok) {
                        a.out2[o] = sv2;
                    }
                }
            }
    }
}

// WM x WN 16x16 MFMA tiles per wave, 2 x 2 waves: block tile (32 WM) x (32 WN).  Smaller tiles are used when the
// 64 x 64 grid would leave most of the 256 CUs idle (the mid-size decoder stages are MFMA-bound per CU).
template <int BK, int WM, int WN, bool UNR = false>
__global__ __launch_bounds__(256, (WM * WN > 12 ? 3 : 4)) void k_conv_gemm(ConvGemmArgs a) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int LD = BK + 8;          // LDS row stride in bf16 elements (+16 B: the 16 rows of a fragment read hit distinct banks)
    constexpr int SEG = BK / 8;         // 8-element segments per row
    constexpr int NA = (BM * SEG + 255) / 256, NB = (BN * SEG + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16_t As[3][BM * LD];     // the h / m / l planes of the activation tile
    __shared__ __attribute__((aligned(16))) bf16_t Bs[BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = (wave >> 1) * 16 * WM, wn = (wave & 1) * 16 * WN;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nck = a.Cin / BK, nit = CG_DEV(a, 1) ? 0 : a.n_taps * nck;

    // staging slots of this thread: (row, segment) pairs of the A and B tiles
    int a_b[NA], a_t[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + 256 * i, am = m0 + idx / SEG;
        a_ok[i] = idx < BM * SEG && am < a.M;
        a_b[i] = a_ok[i] ? am / a.L : 0;
        a_t[i] = a_ok[i] ? am % a.L : 0;
    }
    // the operands of step it+1 are requested before the MFMAs of step it
    float4 v0[NA], v1[NA];
    uint4 wv[NB];
    auto fetch = [&](int it) {
        const int tap = it / nck, c0 = (it - tap * nck) * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int seg = (tid + 256 * i) % SEG;
            v0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            v1[i] = v0[i];
            if (a_ok[i]) {
                const int st = a_t[i] - a.off[tap];      // (off < 0: the tap looks ahead; rows past the request's end read as zero)
                const float* arow = nullptr;
                if (st >= a.L) arow = nullptr;
                else if (st >= 0) arow = a.x + ((size_t)a_b[i] * a.L + st) * a.Cin;
                else if (a.state && a.P + st >= 0) arow = a.state + ((size_t)a.slots[a_b[i]] * a.P + (a.P + st)) * a.Cin;
                if (arow && a.x_s2) {          // (v0, v1) = the 8 h terms and the 8 m terms
                    const bf16_t* r16 = reinterpret_cast<const bf16_t*>(arow);
                    v0[i] = *reinterpret_cast<const float4*>(r16 + c0 + seg * 8);
                    v1[i] = *reinterpret_cast<const float4*>(r16 + a.Cin + c0 + seg * 8);
                } else if (arow) {
                    v0[i] = *reinterpret_cast<const float4*>(arow + c0 + seg * 8);
                    v1[i] = *reinterpret_cast<const float4*>(arow + c0 + seg * 8 + 4);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i, bn = n0 + idx / SEG;
            wv[i] = make_uint4(0, 0, 0, 0);
            if (idx < BN * SEG && bn < a.N)
                wv[i] = *reinterpret_cast<const uint4*>(a.w + ((size_t)tap * a.N + bn) * a.Cin + c0 + (idx % SEG) * 8);
        }
    };
    fetch(0);
    // UNR (few-block launches of the small tiles: one request's chunk): the plane count (1 .. 3 bf16 terms per activation) is a compile-time
    // constant inside the loop.  As a run-time trip count the term loop is not unrolled and every MFMA sits behind its own ds_read + wait —
    // 12 dependent LDS round trips per K step of the 32 x 32 tile; unrolled, a k-block's operand reads go out together (one-request chunk
    // 1.76 -> 1.72 ms).  Many-block launches keep the run-time count: the unrolled reads cost 30 registers, i.e. a wave per SIMD, and
    // there the other waves hide the round trips (8-request chunk 3.34 -> 3.36 ms unrolled).
    auto kloop = [&](auto np_tag) {
    constexpr int NP = decltype(np_tag)::value;
    for (int it = 0; it < nit; ++it) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BM * SEG) {
                const int o = (idx / SEG) * LD + (idx % SEG) * 8;
                if (a.x_s2) {
                    *reinterpret_cast<float4*>(&As[0][o]) = v0[i];
                    *reinterpret_cast<float4*>(&As[1][o]) = v1[i];
                } else {
                    uint4 h, m, l;
                    split3(v0[i], v1[i], h, m, l);
                    *reinterpret_cast<uint4*>(&As[0][o]) = h;
                    if ((NP ? NP : a.planes) > 1) *reinterpret_cast<uint4*>(&As[1][o]) = m;
                    if ((NP ? NP : a.planes) > 2) *reinterpret_cast<uint4*>(&As[2][o]) = l;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BN * SEG) *reinterpret_cast<uint4*>(&Bs[(idx / SEG) * LD + (idx % SEG) * 8]) = wv[i];
        }
        __syncthreads();
        if (it + 1 < nit) fetch(it + 1);
        // lane (fr, g) holds 8 consecutive k of row fr of every 32-wide K block: one 16-byte LDS read per operand
        const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
        for (int kb = 0; kb < BK; kb += 32) {
            uint4 bfr[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(&Bs[(wn + j * 16 + fr) * LD + kb + fk]);
            // (NP = 0: the run-time count — the 128-row tiles run 3 .. 4 blocks per CU on a register budget the unrolled reads do not fit)
#pragma unroll
            for (int t = (NP ? NP : a.planes) - 1; t >= 0; --t) {      // smallest term first
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const uint4 afr = *reinterpret_cast<const uint4*>(&As[t][(wm + i * 16 + fr) * LD + kb + fk]);
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_cbf8(afr), as_cbf8(bfr[j]), acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    };
    if constexpr (!UNR) kloop(std::integral_constant<int, 0>{});
    else if (a.planes >= 3) kloop(std::integral_constant<int, 3>{});
    else if (a.planes == 2) kloop(std::integral_constant<int, 2>{});
    else kloop(std::integral_constant<int, 1>{});
    conv_epilogue<WM, WN>(a, acc, m0, n0, wm, wn, lane);
}

// Multi-tap convolution over S2 rows, activation tile staged ONCE per 32-channel chunk for all taps: the BM + H rows (H = the largest
// look-back) a tile's taps read go to LDS together and every tap's MFMAs read them at its own row shift, so the activation staging
// (global loads, LDS writes, barriers) is paid once per chunk instead of once per tap (7 x for the residual units' dilated convs, whose
// time was that staging: 365 us for 7 taps against 233 us for the whole 1-tap conv2 at 614 k rows).  Weights: one tap's BN x 32 slice at
// a time, double-buffered, one barrier per tap.  Accumulation order: chunk-major, taps inside, the m term before the h term.
// Requires: all offsets in [0, H], L % BM == 0 (a tile lies inside one request).  Dynamic LDS: (2 (BM + H) + 2 BN) x 80 bytes.
template <int WM, int WN>
__device__ __forceinline__ void conv_taps_kloop(const ConvGemmArgs& a, int H, unsigned char* smem_raw, f32x4 (&acc)[WM][WN], int m0, int n0) {
    constexpr int BM = 32 * WM, BN = 32 * WN, LD = 40, NBR = (BN * 4 + 255) / 256;
    const int R = BM + H;
    bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);          // [2][R][LD]
    bf16_t* Bs = As + (size_t)2 * R * LD;                       // [2][BN][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 16 * WM, wn = (wave & 1) * 16 * WN;
    const int rb = m0 / a.L, t0 = m0 - rb * a.L;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t* xrows = reinterpret_cast<const bf16_t*>(a.x) + (size_t)rb * a.L * 2 * a.Cin;
    const bf16_t* srows = a.state ? reinterpret_cast<const bf16_t*>(a.state) + (size_t)a.slots[rb] * a.P * 2 * a.Cin : nullptr;
    // Weights: the (chunk, tap) pairs form ONE stream s = chunk * n_taps + tap; the slice of step s + 2 is requested before the MFMAs of
    // step s and goes to LDS behind the MFMAs of step s + 1 (two register sets, chosen by the parity of s: the stream loop is unrolled by
    // two so that the choice is static).  One step ahead — the round-3 form — a slice had one tap's MFMAs (24 of them, ~0.16 us) to come
    // back from L2 (~0.6 us): every tap waited for its weights, the matrix cores were 27 % busy.  LDS buffers: parity of s, as before.
    uint4 wv0[NBR], wv1[NBR];
    auto fetchB = [&](uint4 (&wv)[NBR], int tap, int c0) {
#pragma unroll
        for (int i = 0; i < NBR; ++i) {
            const int idx = tid + 256 * i, bn = n0 + (idx >> 2);
            wv[i] = make_uint4(0, 0, 0, 0);
            if (idx < BN * 4 && bn < a.N) wv[i] = *reinterpret_cast<const uint4*>(a.w + ((size_t)tap * a.N + bn) * a.Cin + c0 + (idx & 3) * 8);
        }
    };
    auto storeB = [&](const uint4 (&wv)[NBR], int q) {
#pragma unroll
        for (int i = 0; i < NBR; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BN * 4) *reinterpret_cast<uint4*>(Bs + ((size_t)q * BN + (idx >> 2)) * LD + (idx & 3) * 8) = wv[i];
        }
    };
    // the tile's rows of the NEXT chunk travel in registers while this chunk's taps run (all requests issued together, nothing waits
    // on them until the chunk boundary)
    constexpr int NAR = ((BM + 64) * 8 + 255) / 256;
    uint4 av[NAR];
    auto fetchA = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NAR; ++i) {
            const int e = tid + 256 * i, j = e >> 3, p = (e >> 2) & 1, seg = e & 3, st = t0 - H + j;
            av[i] = make_uint4(0, 0, 0, 0);
            const bf16_t* row = nullptr;
            if (e < R * 8) {
                if (st >= 0) row = xrows + (size_t)st * 2 * a.Cin;
                else if (srows && a.P + st >= 0) row = srows + (size_t)(a.P + st) * 2 * a.Cin;
            }
            if (row) av[i] = *reinterpret_cast<const uint4*>(row + p * a.Cin + c0 + seg * 8);
        }
    };
    auto storeA = [&]() {
#pragma unroll
        for (int i = 0; i < NAR; ++i) {
            const int e = tid + 256 * i, j = e >> 3, p = (e >> 2) & 1, seg = e & 3;
            if (e < R * 8) *reinterpret_cast<uint4*>(As + ((size_t)p * R + j) * LD + seg * 8) = av[i];
        }
    };
    const int nck = CG_DEV(a, 1) ? 0 : a.Cin >> 5;
    const int nt = a.n_taps, total = nck * nt;
    if (total == 0) return;
    // (tap, chunk offset) of stream index s
    int f_tap = 0, f_c0 = 0;                       // of the next slice to REQUEST
    auto advance_f = [&]() { if (++f_tap == nt) { f_tap = 0; f_c0 += 32; } };
    fetchA(0);
    fetchB(wv0, 0, 0); advance_f();
    if (total > 1) { fetchB(wv1, f_tap, f_c0); advance_f(); }
    __syncthreads();              // (the caller may have used the LDS before)
    storeB(wv0, 0);
    int tap = 0, ck = 0;                            // of the step being computed
    // one step: F = the register set that is free (its slice, step s, is in LDS), S = the set holding step s + 1
    auto step = [&](int s_, uint4 (&F)[NBR], const uint4 (&S)[NBR]) {
        if (tap == 0) {           // chunk boundary: the previous step's closing barrier says nobody reads the old tile any more
            storeA();
            __syncthreads();
            if (ck + 1 < nck) fetchA((ck + 1) * 32);
        }
        if (s_ + 2 < total) { fetchB(F, f_tap, f_c0); advance_f(); }
        const int sh = H - a.off[tap];
        const bf16_t* Bq = Bs + (size_t)(s_ & 1) * BN * LD;
        uint4 bfr[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(Bq + (wn + j * 16 + fr) * LD + fk);
#pragma unroll
        for (int p = 1; p >= 0; --p) {      // smaller term first
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const uint4 afr = *reinterpret_cast<const uint4*>(As + ((size_t)p * R + wm + i * 16 + fr + sh) * LD + fk);
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_cbf8(afr), as_cbf8(bfr[j]), acc[i][j], 0, 0, 0);
            }
        }
        if (s_ + 1 < total) storeB(S, (s_ + 1) & 1);
        __syncthreads();          // the next step's weights are visible; this step's buffer may be overwritten by the step after next
        if (++tap == nt) { tap = 0; ++ck; }
    };
    for (int s_ = 0; s_ < total; s_ += 2) {
        step(s_, wv0, wv1);
        if (s_ + 1 < total) step(s_ + 1, wv1, wv0);
    }
}

template <int WM, int WN>
__global__ __launch_bounds__(256, (WM * WN >= 12 ? 3 : 4)) void k_conv_taps(ConvGemmArgs a, int H) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = (wave >> 1) * 16 * WM, wn = (wave & 1) * 16 * WN;
    f32x4 acc[WM][WN];
    conv_taps_kloop<WM, WN>(a, H, smem_raw, acc, m0, n0);
    conv_epilogue<WM, WN>(a, acc, m0, n0, wm, wn, lane);
}

// A whole residual unit of the decoder in ONE kernel:  h += conv2(act2(conv1(x)))  and  out2 = act_next(h)  — conv1 = the dilated 7-tap
// conv over S2 rows (k_conv_taps' K loop), conv2 = the 1-tap C -> C conv.  The block's tile covers ALL C output channels of conv1, so the
// activated intermediate (the whole K of conv2) is on chip: it goes from the accumulators through Snake + the two-term split into LDS
// as the A operand of the second GEMM instead of to HBM and back (per unit at 614 k rows x 96 channels: 236 MB written + 236 MB read).
// Accumulation orders are those of the two kernels it replaces (conv2: 32-channel chunks in order, the m term before the h term), so the
// results are bit-identical.  a2.out2 must not alias a1.x: a neighbouring tile still reads its halo rows of x.
// Dynamic LDS: max((2 (BM + H) + 2 BN) x 80, 2 BM (C + 8) x 2 + 2 BN x 80) bytes.
template <int WM, int WN>
__global__ __launch_bounds__(256, (WM * WN >= 12 ? 2 : 3)) void k_res_unit(ConvGemmArgs a1, ConvGemmArgs a2, int H) {
    constexpr int BM = 32 * WM, BN = 32 * WN, LD = 40, NBR = (BN * 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM;
    const int wm = (wave >> 1) * 16 * WM, wn = (wave & 1) * 16 * WN;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    f32x4 acc[WM][WN];
    conv_taps_kloop<WM, WN>(a1, H, smem_raw, acc, m0, 0);
    // ---- the activated intermediate as the second GEMM's A operand: [2 terms][BM][C + 8] bf16
    const int C = a1.N, LD2 = C + 8;
    bf16_t* A2 = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* B2 = A2 + (size_t)2 * BM * LD2;                     // [2][BN][LD]
    uint4 wv[NBR];
    auto fetchB = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NBR; ++i) {
            const int idx = tid + 256 * i, bn = idx >> 2;
            wv[i] = make_uint4(0, 0, 0, 0);
            if (idx < BN * 4 && bn < a2.N) wv[i] = *reinterpret_cast<const uint4*>(a2.w + (size_t)bn * a2.Cin + c0 + (idx & 3) * 8);
        }
    };
    auto storeB = [&](int q) {
#pragma unroll
        for (int i = 0; i < NBR; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BN * 4) *reinterpret_cast<uint4*>(B2 + ((size_t)q * BN + (idx >> 2)) * LD + (idx & 3) * 8) = wv[i];
        }
    };
    fetchB(0);
    __syncthreads();                                            // every wave is done with the first GEMM's tiles
    {
        const int l15 = lane & 15, rg = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = wn + j * 16 + l15;
            const float cb = a1.bias ? a1.bias[n % a1.bias_mod] : 0.0f;
            const int cc = n % a1.sn_mod;
            const float ca = a1.sn_alpha[cc], ci = a1.sn_invb[cc];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sv = snake_f(acc[i][j][r] + cb, ca, ci);
                    bf16_t hb, mb;
                    split2(sv, hb, mb);
                    const int row = wm + i * 16 + rg + r;
                    A2[(size_t)row * LD2 + n] = hb;
                    A2[((size_t)BM + row) * LD2 + n] = mb;
                    acc[i][j][r] = 0.0f;
                }
        }
    }
    storeB(0);
    __syncthreads();
    const int nck = C >> 5;
    for (int ck = 0; ck < nck; ++ck) {
        if (ck + 1 < nck) fetchB((ck + 1) * 32);
        const bf16_t* Bq = B2 + (size_t)(ck & 1) * BN * LD;
        uint4 bfr[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(Bq + (wn + j * 16 + fr) * LD + fk);
#pragma unroll
        for (int p = 1; p >= 0; --p) {          // smaller term first
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const uint4 afr = *reinterpret_cast<const uint4*>(A2 + ((size_t)p * BM + wm + i * 16 + fr) * LD2 + ck * 32 + fk);
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_cbf8(afr), as_cbf8(bfr[j]), acc[i][j], 0, 0, 0);
            }
        }
        if (ck + 1 < nck) storeB((ck + 1) & 1);
        __syncthreads();
    }
    conv_epilogue<WM, WN>(a2, acc, m0, 0, wm, wn, lane);
}

// LayerNorm row statistics by the 16 lanes l16 = 0..15 of a row group (two passes over float4 chunks, butterfly inside the group).
// Shared by the GEMM with the fused LayerNorm and by k_flow_ln, so a row normalises to the same bits on either path (a request's output
// must not depend on how many requests share its batch).
__device__ __forceinline__ void ln_row_stats(const float* row, int C, int l16, float eps, float& mean, float& rstd) {
    const float4* xr = reinterpret_cast<const float4*>(row);
    float sm = 0.0f;
    for (int c = l16; c < C / 4; c += 16) { const float4 v = xr[c]; sm += (v.x + v.y) + (v.z + v.w); }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) sm += __shfl_xor(sm, off, 64);
    mean = sm / (float)C;
    float vs = 0.0f;
    for (int c = l16; c < C / 4; c += 16) {
        const float4 v = xr[c];
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        vs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) vs += __shfl_xor(vs, off, 64);
    rstd = rsqrtf(vs / (float)C + eps);
}

// Few-row variant (M <= 48: the transformer and the first upsampling stages of a single request's chunk): tile
// 16(M) x 64(N), one n-tile per wave, K walked in BK-wide chunks (128 where Cin allows) with the same one-step
// register prefetch.  These stages run a handful of blocks: time = number of dependent K steps, so the steps are wide.
template <int BK, bool LN = false>
__global__ __launch_bounds__(256) void k_conv_gemm_skinny(ConvGemmArgs a) {
    constexpr int LD = BK + 8;                      // bf16 elements
    constexpr int NA = (16 * BK / 4 + 255) / 256;   // float4 per thread (activations)
    constexpr int NB = 64 * BK / 8 / 256;           // uint4 per thread (weights)
    constexpr int SEGA = BK / 4, SEGB = BK / 8;
    __shared__ __attribute__((aligned(16))) bf16_t As[3][16 * LD];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nck = a.Cin / BK, nit = a.n_taps * nck;
    float4 av[NA];
    uint4 bv[NB];
    auto fetch = [&](int it) {
        const int tap = it / nck, c0 = (it - tap * nck) * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i, r = idx / SEGA, seg = idx % SEGA;
            av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int am = m0 + r;
            if (idx < 16 * SEGA && am < a.M) {
                const int ab = am / a.L, st = am % a.L - a.off[tap];
                const float* arow = nullptr;
                if (st >= a.L) arow = nullptr;
                else if (st >= 0) arow = a.x + ((size_t)ab * a.L + st) * a.Cin;
                else if (a.state && a.P + st >= 0) arow = a.state + ((size_t)a.slots[ab] * a.P + (a.P + st)) * a.Cin;
                if (arow && a.x_s2) {          // (xy, zw) = the 4 h terms and the 4 m terms
                    const bf16_t* r16 = reinterpret_cast<const bf16_t*>(arow);
                    const float2 h2 = *reinterpret_cast<const float2*>(r16 + c0 + seg * 4);
                    const float2 m2 = *reinterpret_cast<const float2*>(r16 + a.Cin + c0 + seg * 4);
                    av[i] = make_float4(h2.x, h2.y, m2.x, m2.y);
                } else if (arow) {
                    av[i] = *reinterpret_cast<const float4*>(arow + c0 + seg * 4);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i, r = idx / SEGB, seg = idx % SEGB;
            const int bn = n0 + r;
            bv[i] = make_uint4(0, 0, 0, 0);
            if (bn < a.N) bv[i] = *reinterpret_cast<const uint4*>(a.w + ((size_t)tap * a.N + bn) * a.Cin + c0 + seg * 8);
        }
    };
    fetch(0);
    __shared__ float ln_stat[LN ? 32 : 1];
    if (LN) {      // row statistics of the block's 16 rows (two passes, 16 lanes per row), while the first operands are in flight
        const int r = tid >> 4, l16 = tid & 15, am = m0 + r;
        float mean = 0.0f, rstd = 0.0f;
        if (am < a.M) ln_row_stats(a.x + (size_t)am * a.Cin, a.Cin, l16, a.ln_eps, mean, rstd);
        if (l16 == 0) { ln_stat[2 * r] = mean; ln_stat[2 * r + 1] = rstd; }
    }
    auto kloop = [&](auto np_tag) {       // (the plane count as a compile-time constant: see k_conv_gemm)
    constexpr int NP = decltype(np_tag)::value;
    for (int it = 0; it < nit; ++it) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 16 * SEGA && a.x_s2 && !LN) {
                const int o = (idx / SEGA) * LD + (idx % SEGA) * 4;
                *reinterpret_cast<float2*>(&As[0][o]) = make_float2(av[i].x, av[i].y);
                *reinterpret_cast<float2*>(&As[1][o]) = make_float2(av[i].z, av[i].w);
            } else if (idx < 16 * SEGA) {      // 4 fp32 -> 4 bf16 per plane (8 bytes)
                float x4[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
                if (LN) {
                    const int r = idx / SEGA, c = (it % nck) * BK + (idx % SEGA) * 4;
                    const float mean = ln_stat[2 * r], rstd = ln_stat[2 * r + 1];
                    const float4 w4 = *reinterpret_cast<const float4*>(a.ln_w + c), b4 = *reinterpret_cast<const float4*>(a.ln_b + c);
                    x4[0] = (x4[0] - mean) * rstd * w4.x + b4.x; x4[1] = (x4[1] - mean) * rstd * w4.y + b4.y;
                    x4[2] = (x4[2] - mean) * rstd * w4.z + b4.z; x4[3] = (x4[3] - mean) * rstd * w4.w + b4.w;
                }
                unsigned short hb[4], mb[4], lb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hb[e] = f2bf(x4[e]);
                    const float r1 = x4[e] - bf2f(hb[e]);
                    mb[e] = f2bf(r1);
                    lb[e] = f2bf(r1 - bf2f(mb[e]));
                }
                const int o = (idx / SEGA) * LD + (idx % SEGA) * 4;
                *reinterpret_cast<uint2*>(&As[0][o]) = make_uint2(hb[0] | (unsigned)hb[1] << 16, hb[2] | (unsigned)hb[3] << 16);
                *reinterpret_cast<uint2*>(&As[1][o]) = make_uint2(mb[0] | (unsigned)mb[1] << 16, mb[2] | (unsigned)mb[3] << 16);
                *reinterpret_cast<uint2*>(&As[2][o]) = make_uint2(lb[0] | (unsigned)lb[1] << 16, lb[2] | (unsigned)lb[3] << 16);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<uint4*>(&Bs[(idx / SEGB) * LD + (idx % SEGB) * 8]) = bv[i];
        }
        __syncthreads();
        if (it + 1 < nit) fetch(it + 1);
        const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
        for (int kb = 0; kb < BK; kb += 32) {
            const uint4 bfr = *reinterpret_cast<const uint4*>(&Bs[(wave * 16 + fr) * LD + kb + fk]);
#pragma unroll
            for (int t = NP - 1; t >= 0; --t) {
                const uint4 afr = *reinterpret_cast<const uint4*>(&As[t][fr * LD + kb + fk]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_cbf8(afr), as_cbf8(bfr), acc, 0, 0, 0);
            }
        }
    }
    };
    if (a.planes >= 3) kloop(std::integral_constant<int, 3>{});
    else if (a.planes == 2) kloop(std::integral_constant<int, 2>{});
    else kloop(std::integral_constant<int, 1>{});
    const int n = n0 + wave * 16 + (lane & 15);
    if (n >= a.N) return;
    const float bvv = a.bias ? a.bias[n % a.bias_mod] : 0.0f;
    const float sv = a.scale ? a.scale[n] : 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + (lane >> 4) * 4 + r;
        if (m >= a.M) continue;
        float v = acc[r] + bvv;
        if (a.gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (a.rscale) v = a.rscale[m] * v;
        const size_t o = (size_t)m * a.N + n;
        if (a.res) v = a.res[o] + sv * v;
        else if (a.scale) v = sv * v;
        if (a.out) a.out[o] = v;
        if (a.out2) {
            const float sv2 = snake_f(v, a.sn_alpha[n % a.sn_mod], a.sn_invb[n % a.sn_mod]);
            if (a.out2_s2) {
                const int cw = a.sn_mod, j = n / cw, c = n - j * cw;
                bf16_t* r16 = reinterpret_cast<bf16_t*>(a.out2 + (size_t)m * a.N) + (size_t)j * 2 * cw;
                split2(sv2, r16[c], r16[cw + c]);
            } else {
                a.out2[o] = sv2;
            }
        }
    }
}

// Few-row GEMM with no dependent K steps ("rows" kernel; the flow detokenizers' 56..2752-row linears and causal convs: thousands of
// launches per chunk whose time is exposed memory latency, not arithmetic).  A block owns 16 MT rows x 16 NT columns; its WV waves split
// the n_taps * Cin reduction into WV runs of KB 32-wide k-blocks and each lane requests ALL its operands up front, straight in the
// layout an MFMA operand register holds (A: 8 consecutive fp32 of row lane % 16; B: 8 consecutive bf16 of weight row lane % 16) — one
// exposed round trip per launch, no LDS staging, no barrier until the cross-wave sum (fixed order w0 + w1 + ... ).  With the fused
// LayerNorm the row statistics come from the operand registers while the weights are in flight.  A row's result depends on that row only
// and on (K, WV) — never on MT / NT, which the launcher picks from the row count: more rows per block when the call is large enough
// that the operand traffic from L2 (12 KB per 16 x 16 output tile at MT = 1, NT = 4, K = 256) is what bounds it.  WV is a function of K
// alone (8 waves from K = 512 on: half the serial work per wave), so a request's output does not depend on what shares its batch.
template <int KB, int NT, bool LN, int MT, int WV>
__global__ __launch_bounds__(64 * WV) void k_rows_gemm(ConvGemmArgs a) {
    __shared__ float red[WV][MT * NT][256];
    __shared__ float ln_stat[LN ? 2 * WV * 16 * MT : 1];
    // (the wave index provably uniform: the tap of a k-block, and with it a.off[tap], is then a scalar — read per lane from the argument
    // block it was a dependent vector load, one round trip in front of every operand row's address)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 16 * MT, n0 = blockIdx.x * 16 * NT;
    const int kpt = a.Cin >> 5;                     // k-blocks per tap
    float4 av[MT][KB][2];
    uint4 bv[NT][KB];
    float4 lw[LN ? KB : 1][2], lb[LN ? KB : 1][2];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        const int kbi = wave * KB + j, tap = kbi / kpt, c0 = (kbi - tap * kpt) * 32 + kq * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int am = m0 + mt * 16 + fr;
            const float* arow = nullptr;
            if (am < a.M) {
                const int ab = am / a.L, st = am % a.L - a.off[tap];
                if (st >= a.L) arow = nullptr;
                else if (st >= 0) arow = a.x + ((size_t)ab * a.L + st) * a.Cin;
                else if (a.state && a.P + st >= 0) arow = a.state + ((size_t)a.slots[ab] * a.P + (a.P + st)) * a.Cin;
            }
            av[mt][j][0] = av[mt][j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow) {
                av[mt][j][0] = *reinterpret_cast<const float4*>(arow + c0);
                av[mt][j][1] = *reinterpret_cast<const float4*>(arow + c0 + 4);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int bn = n0 + t * 16 + fr;
            bv[t][j] = make_uint4(0, 0, 0, 0);
            if (bn < a.N) bv[t][j] = *reinterpret_cast<const uint4*>(a.w + ((size_t)tap * a.N + bn) * a.Cin + c0);
        }
        if (LN) {
            lw[j][0] = *reinterpret_cast<const float4*>(a.ln_w + c0); lw[j][1] = *reinterpret_cast<const float4*>(a.ln_w + c0 + 4);
            lb[j][0] = *reinterpret_cast<const float4*>(a.ln_b + c0); lb[j][1] = *reinterpret_cast<const float4*>(a.ln_b + c0 + 4);
        }
    }
    float mean[MT], rstd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) mean[mt] = rstd[mt] = 0.0f;
    if (LN) {
        // Row statistics from the operand registers (one tap: the block's WV KB k-blocks are exactly the row): lane (fr, kq) of wave w
        // holds 8 KB values of row fr; two passes (mean, then centred squares), each = lane sum -> the row's four k groups by two
        // butterfly steps -> the waves in the fixed order w0 + w1 + ...  No second trip to memory for the centred pass.
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < KB; ++j)
                s += ((av[mt][j][0].x + av[mt][j][0].y) + (av[mt][j][0].z + av[mt][j][0].w)) + ((av[mt][j][1].x + av[mt][j][1].y) + (av[mt][j][1].z + av[mt][j][1].w));
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if (kq == 0) ln_stat[(mt * WV + wave) * 16 + fr] = s;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float tot = ln_stat[mt * WV * 16 + fr];
#pragma unroll
            for (int w = 1; w < WV; ++w) tot += ln_stat[(mt * WV + w) * 16 + fr];
            mean[mt] = tot / (float)a.Cin;
            float vs = 0.0f;
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const float d0 = av[mt][j][0].x - mean[mt], d1 = av[mt][j][0].y - mean[mt], d2 = av[mt][j][0].z - mean[mt], d3 = av[mt][j][0].w - mean[mt];
                const float d4 = av[mt][j][1].x - mean[mt], d5 = av[mt][j][1].y - mean[mt], d6 = av[mt][j][1].z - mean[mt], d7 = av[mt][j][1].w - mean[mt];
                vs += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
            }
            vs += __shfl_xor(vs, 16, 64); vs += __shfl_xor(vs, 32, 64);
            if (kq == 0) ln_stat[(MT * WV + mt * WV + wave) * 16 + fr] = vs;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float tot = ln_stat[(MT * WV + mt * WV) * 16 + fr];
#pragma unroll
            for (int w = 1; w < WV; ++w) tot += ln_stat[(MT * WV + mt * WV + w) * 16 + fr];
            rstd[mt] = rsqrtf(tot / (float)a.Cin + a.ln_eps);
        }
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KB; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float x8[8] = {av[mt][j][0].x, av[mt][j][0].y, av[mt][j][0].z, av[mt][j][0].w, av[mt][j][1].x, av[mt][j][1].y, av[mt][j][1].z, av[mt][j][1].w};
            if (LN) {
                const float w8[8] = {lw[j][0].x, lw[j][0].y, lw[j][0].z, lw[j][0].w, lw[j][1].x, lw[j][1].y, lw[j][1].z, lw[j][1].w};
                const float b8[8] = {lb[j][0].x, lb[j][0].y, lb[j][0].z, lb[j][0].w, lb[j][1].x, lb[j][1].y, lb[j][1].z, lb[j][1].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) x8[e] = (x8[e] - mean[mt]) * rstd[mt] * w8[e] + b8[e];
            }
            unsigned short hb[8], mb[8], lbb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hb[e] = f2bf(x8[e]);
                const float r1 = x8[e] - bf2f(hb[e]);
                mb[e] = f2bf(r1);
                lbb[e] = f2bf(r1 - bf2f(mb[e]));
            }
            const uint4 ah = make_uint4(hb[0] | (unsigned)hb[1] << 16, hb[2] | (unsigned)hb[3] << 16, hb[4] | (unsigned)hb[5] << 16, hb[6] | (unsigned)hb[7] << 16);
            const uint4 amid = make_uint4(mb[0] | (unsigned)mb[1] << 16, mb[2] | (unsigned)mb[3] << 16, mb[4] | (unsigned)mb[5] << 16, mb[6] | (unsigned)mb[7] << 16);
            const uint4 al = make_uint4(lbb[0] | (unsigned)lbb[1] << 16, lbb[2] | (unsigned)lbb[3] << 16, lbb[4] | (unsigned)lbb[5] << 16, lbb[6] | (unsigned)lbb[7] << 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {      // plane order of the staged kernels: low terms first
                if (a.planes >= 3) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_cbf8(al), as_cbf8(bv[t][j]), acc[mt][t], 0, 0, 0);
                if (a.planes >= 2) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_cbf8(amid), as_cbf8(bv[t][j]), acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_cbf8(ah), as_cbf8(bv[t][j]), acc[mt][t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][mt * NT + t][r * 64 + lane] = acc[mt][t][r];
    __syncthreads();
    // work item e = (tile, accumulator component r): row m0 + 16 mt + 4 (lane / 16) + r, column n0 + 16 t + lane % 16; wave w takes
    // the items w, w + WV, ...  Two passes: every operand of every item of this wave (bias, scale, row scale, residual, Snake constants)
    // is requested first, branch-free (clamped indices), THEN the arithmetic and the stores — item by item the loads of item i + 1 could not
    // move above the stores of item i (the residual may alias the output), and an item cost up to three dependent round trips: the
    // epilogue was most of this kernel's time.  Per element the operations and their order are unchanged.
    constexpr int NE = (MT * NT * 4 + WV - 1) / WV;
    float e_b[NE], e_s[NE], e_rs[NE], e_res[NE], e_a[NE], e_i[NE];
    const int Mc = a.M - 1, Nc = a.N - 1;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = wave + i * WV, tile = (e >> 2) < MT * NT ? (e >> 2) : 0, r = e & 3, mt = tile / NT, t = tile - mt * NT;
        int m = m0 + mt * 16 + kq * 4 + r, n = n0 + t * 16 + fr;
        m = m < Mc ? m : Mc; n = n < Nc ? n : Nc;
        e_b[i] = a.bias ? a.bias[n % a.bias_mod] : 0.0f;
        e_s[i] = a.scale ? a.scale[n] : 1.0f;
        e_rs[i] = a.rscale ? a.rscale[m] : 1.0f;
        e_res[i] = a.res ? a.res[(size_t)m * a.N + n] : 0.0f;
        e_a[i] = a.out2 ? a.sn_alpha[n % a.sn_mod] : 0.0f;
        e_i[i] = a.out2 ? a.sn_invb[n % a.sn_mod] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = wave + i * WV;
        if (e >= MT * NT * 4) continue;
        const int tile = e >> 2, r = e & 3, mt = tile / NT, t = tile - mt * NT;
        const int m = m0 + mt * 16 + kq * 4 + r, n = n0 + t * 16 + fr;
        if (m >= a.M || n >= a.N) continue;
        const int idx = r * 64 + lane;
        float v = red[0][tile][idx];
#pragma unroll
        for (int w = 1; w < WV; ++w) v += red[w][tile][idx];
        v += e_b[i];
        const float sv = e_s[i];
        if (a.gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (a.rscale) v = e_rs[i] * v;
        const size_t o = (size_t)m * a.N + n;
        if (a.res) v = e_res[i] + sv * v;
        else if (a.scale) v = sv * v;
        if (a.out) a.out[o] = v;
        if (a.out2) {
            const float sv2 = snake_f(v, e_a[i], e_i[i]);
            if (a.out2_s2) {
                const int cw = a.sn_mod, jj = n / cw, c = n - jj * cw;
                bf16_t* r16 = reinterpret_cast<bf16_t*>(a.out2 + (size_t)m * a.N) + (size_t)jj * 2 * cw;
                split2(sv2, r16[c], r16[cw + c]);
            } else {
                a.out2[o] = sv2;
            }
        }
    }
}

// ================================================================================================
// small kernels
// ================================================================================================
// RVQ decode: q0 = emb[0][code0], qr = sum_{k>=1} emb[k][code_k] (sequential in k)      qwen3_codec.py:1204-1210
__global__ __launch_bounds__(256) void k_rvq(const int* codes, int code_stride, const float* emb, int Q, int bins, int vq,
                                             float* q0, float* qr) {
    const int row = blockIdx.x;
    const int* cr = codes + (size_t)row * code_stride;
    for (int d = threadIdx.x; d < vq; d += 256) {
        int c0 = cr[0];
        c0 = c0 < 0 ? 0 : (c0 >= bins ? bins - 1 : c0);          // postprocess clamps to [0, bins-1] (qwen3_tts.py:2027)
        q0[(size_t)row * vq + d] = emb[((size_t)c0) * vq + d];
        float s = 0.0f;
        for (int k = 1; k < Q; ++k) {
            int c = cr[k];
            c = c < 0 ? 0 : (c >= bins ? bins - 1 : c);
            s = s + emb[((size_t)k * bins + c) * vq + d];
        }
        qr[(size_t)row * vq + d] = s;
    }
}

// new_state = last P rows of [state ++ x]; one thread per (request, channel), ascending rows (in-place safe)
__global__ __launch_bounds__(256) void k_state_update(float* state, const int* slots, const float* x, int L, int P, int C) {
    const int b = blockIdx.y;
    float* s = state + (size_t)slots[b] * P * C;
    const float* xb = x + (size_t)b * L * C;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256)
        for (int j = 0; j < P; ++j) {
            const int e = L + j;   // index into ext = [state(P) ++ x(L)]
            s[(size_t)j * C + c] = e < P ? s[(size_t)e * C + c] : xb[(size_t)(e - P) * C + c];
        }
}

// SnakeBeta: y = x + inv_beta[c] * sin(x * alpha[c])^2          qwen3_codec.py:1004-1018 (alpha/beta pre-exponentiated)
__global__ __launch_bounds__(256) void k_snake(const float* x, const float* alpha, const float* inv_beta, float* y,
                                               size_t total, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float v = x[i];
        const float s = sinf(v * alpha[c]);
        y[i] = v + inv_beta[c] * (s * s);
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// codec RMSNorm: y = w * (x * rsqrt(mean(x^2) + eps))       qwen3_codec.py:713-718
__global__ __launch_bounds__(256) void k_rmsnorm_f32(const float* x, const float* w, float* y, int C, float eps) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * C;
    float s = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) s += xr[c] * xr[c];
    s = block_sum(s, red);
    const float r = rsqrtf(s / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) y[(size_t)blockIdx.x * C + c] = w[c] * (xr[c] * r);
}

// RoPE (NeoX halves) on q in place; k rotated and v written to the bf16 KV ring.   qwen3_codec.py:212-236, 614-631
// qkv [rows, 3*HD]; ring [slots][Wn][2][HD] bf16; pos[slot] = tokens already in the window's history.
__global__ __launch_bounds__(256) void k_codec_rope_kv(float* qkv, bf16_t* ring, const int* slots, const long* pos, int T,
                                                       int H, int D, int Wn, const float* inv_freq) {
    const int row = blockIdx.x, b = row / T, t = row % T;
    const int HD = H * D, half = D / 2;
    const long p = pos[slots[b]] + t;
    float* q = qkv + (size_t)row * 3 * HD;
    float* k = q + HD;
    const float* v = q + 2 * HD;
    bf16_t* rk = ring + (((size_t)slots[b] * Wn + (p % Wn)) * 2) * HD;
    bf16_t* rv = rk + HD;
    for (int e = threadIdx.x; e < H * half; e += 256) {
        const int h = e / half, i = e % half;
        const float ang = (float)p * inv_freq[i];
        const float c = cosf(ang), s = sinf(ang);
        const int ia = h * D + i, ib = ia + half;
        const float qa = q[ia], qb = q[ib], ka = k[ia], kb = k[ib];
        q[ia] = qa * c - qb * s;
        q[ib] = qb * c + qa * s;
        rk[ia] = f2bf(ka * c - kb * s);
        rk[ib] = f2bf(kb * c + ka * s);
    }
    for (int e = threadIdx.x; e < HD; e += 256) rv[e] = f2bf(v[e]);
}

// windowed attention over the ring.  Query i (absolute position p0+i) sees absolute positions
// [p0+T-Wn, p0+i]; positions < 0 are the reference's never-written zero slots, which stay visible.
// One block per (head, request): K/V ring tile in LDS (row stride D+1: conflict-free both for "lane = key" in the
// scores and "lane = dim" in PV), the T queries of the chunk spread over the four waves.
__global__ __launch_bounds__(256) void k_codec_attn(const float* qkv, const bf16_t* ring, const int* slots, const long* pos,
                                                    float* out, int T, int H, int D, int Wn) {
    extern __shared__ float sm[];   // K [Wn][D+1], V [Wn][D+1], Q [T][D], P [4][Wn]
    const int LD = D + 1;
    float* Ks = sm;
    float* Vs = Ks + (size_t)Wn * LD;
    float* Qs = Vs + (size_t)Wn * LD;
    float* Ps = Qs + (size_t)T * D;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HD = H * D;
    const int slot = slots[b];
    const long p0 = pos[slot];
    const long lo = p0 + T - Wn;   // absolute position of logical slot 0
    for (int e = tid; e < Wn * D; e += 256) {
        const int j = e / D, d = e % D;
        const long ap = lo + j;
        float kv = 0.0f, vv = 0.0f;
        if (ap >= 0) {
            const bf16_t* r = ring + (((size_t)slot * Wn + (ap % Wn)) * 2) * HD + (size_t)h * D + d;
            kv = bf2f(r[0]);
            vv = bf2f(r[HD]);
        }
        Ks[j * LD + d] = kv;
        Vs[j * LD + d] = vv;
    }
    for (int e = tid; e < T * D; e += 256) Qs[e] = qkv[((size_t)(b * T + e / D)) * 3 * HD + (size_t)h * D + e % D];
    __syncthreads();
    const float scale = rsqrtf((float)D);
    float* P = Ps + (size_t)wave * Wn;
    for (int i = wave; i < T; i += 4) {
        const float* q = Qs + (size_t)i * D;
        const int nvis = Wn - T + i + 1;
        float mx = -INFINITY;
        for (int j = lane; j < Wn; j += 64) {
            float sc = -INFINITY;
            if (j < nvis) {
                sc = 0.0f;
                for (int d = 0; d < D; ++d) sc = fmaf(q[d], Ks[j * LD + d], sc);
                sc *= scale;
            }
            P[j] = sc;
            mx = fmaxf(mx, sc);
        }
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float ls = 0.0f;
        for (int j = lane; j < Wn; j += 64) {
            const float pj = j < nvis ? expf(P[j] - mx) : 0.0f;
            P[j] = pj;
            ls += pj;
        }
        for (int off = 32; off >= 1; off >>= 1) ls += __shfl_xor(ls, off, 64);
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (int d = lane; d < D; d += 64) {
            float o = 0.0f;
            for (int j = 0; j < nvis; ++j) o = fmaf(P[j], Vs[j * LD + d], o);
            out[((size_t)(b * T + i)) * HD + (size_t)h * D + d] = o / ls;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}

// k_codec_rope_kv + k_codec_attn as ONE launch per layer, head dim as a compile-time constant: block (head, request) rotates its
// head's T new q / k rows (q stays in LDS), writes the new K / V rows of its head to the ring and keeps their bf16 values in the LDS
// tile next to the history rows (16-byte ring loads, one 64-bit modulo per row chunk instead of one per element); the T x Wn scores are
// spread over all 256 threads as (query, key) pairs — the old form gave a wave one query at a time and left 56 lanes idle in the second
// key pass — and the per-query softmax + P.V keep their wave-per-query form.  Every output's arithmetic (the fma chain over d of a
// score, the lane sums and butterfly of the softmax, the fma chain over the keys of P.V, the bf16 rounding of the ring rows) is the
// chain of the two kernels above: bit-identical (waveform digests equal).  Chunk of 1 / 8 / 32 requests: 1.95 -> 1.76, 3.53 -> 3.34, 7.70 -> 7.47 ms
// (profiles/round4_codec_attn_ab.txt).
template <int D>
__global__ __launch_bounds__(256) void k_codec_attn2(const float* qkv, bf16_t* ring, const int* slots, const long* pos, float* out, int T,
                                                     int H, int Wn, const float* inv_freq) {
    extern __shared__ float sm[];   // K [Wn][D+1], V [Wn][D+1], Q [T][D], P [T][Wn]
    constexpr int LD = D + 1, half = D / 2, CH = D / 8;
    float* Ks = sm;
    float* Vs = Ks + (size_t)Wn * LD;
    float* Qs = Vs + (size_t)Wn * LD;
    float* Ps = Qs + (size_t)T * D;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HD = H * D;
    const int slot = slots[b];
    const long p0 = pos[slot];
    const long lo = p0 + T - Wn;   // absolute position of logical slot 0
    bf16_t* rbase = ring + (size_t)slot * Wn * 2 * HD + (size_t)h * D;
    // history rows: logical slots 0 .. Wn - T - 1 (absolute positions lo .. p0 - 1; negative = the never-written zero rows)
    for (int c = tid; c < (Wn - T) * CH; c += 256) {
        const int j = c / CH, d0 = (c % CH) * 8;
        const long ap = lo + j;
        uint4 kq = make_uint4(0, 0, 0, 0), vq = kq;
        if (ap >= 0) {
            const bf16_t* r = rbase + (size_t)(ap % Wn) * 2 * HD + d0;
            kq = *reinterpret_cast<const uint4*>(r);
            vq = *reinterpret_cast<const uint4*>(r + HD);
        }
        float* kd = Ks + j * LD + d0;
        float* vd = Vs + j * LD + d0;
        kd[0] = bf2f((bf16_t)(kq.x & 0xffff)); kd[1] = bf2f((bf16_t)(kq.x >> 16)); kd[2] = bf2f((bf16_t)(kq.y & 0xffff)); kd[3] = bf2f((bf16_t)(kq.y >> 16));
        kd[4] = bf2f((bf16_t)(kq.z & 0xffff)); kd[5] = bf2f((bf16_t)(kq.z >> 16)); kd[6] = bf2f((bf16_t)(kq.w & 0xffff)); kd[7] = bf2f((bf16_t)(kq.w >> 16));
        vd[0] = bf2f((bf16_t)(vq.x & 0xffff)); vd[1] = bf2f((bf16_t)(vq.x >> 16)); vd[2] = bf2f((bf16_t)(vq.y & 0xffff)); vd[3] = bf2f((bf16_t)(vq.y >> 16));
        vd[4] = bf2f((bf16_t)(vq.z & 0xffff)); vd[5] = bf2f((bf16_t)(vq.z >> 16)); vd[6] = bf2f((bf16_t)(vq.w & 0xffff)); vd[7] = bf2f((bf16_t)(vq.w >> 16));
    }
    // the chunk's T new rows: RoPE (NeoX halves) on q and k, k / v rounded to bf16 for the ring — and for this launch's own tile
    for (int e = tid; e < T * half; e += 256) {
        const int t = e / half, i = e % half, j = Wn - T + t;
        const long p = p0 + t;
        const float ang = (float)p * inv_freq[i];
        const float c = cosf(ang), s = sinf(ang);
        const float* q = qkv + ((size_t)(b * T + t)) * 3 * HD + (size_t)h * D;
        const float* k = q + HD;
        const float qa = q[i], qb = q[i + half], ka = k[i], kb = k[i + half];
        Qs[t * D + i] = qa * c - qb * s;
        Qs[t * D + i + half] = qb * c + qa * s;
        const bf16_t k0 = f2bf(ka * c - kb * s), k1 = f2bf(kb * c + ka * s);
        bf16_t* rk = rbase + (size_t)(p % Wn) * 2 * HD;
        rk[i] = k0;
        rk[i + half] = k1;
        Ks[j * LD + i] = bf2f(k0);
        Ks[j * LD + i + half] = bf2f(k1);
    }
    for (int e = tid; e < T * D; e += 256) {
        const int t = e / D, d = e % D, j = Wn - T + t;
        const long p = p0 + t;
        const bf16_t v16 = f2bf(qkv[((size_t)(b * T + t)) * 3 * HD + 2 * HD + (size_t)h * D + d]);
        rbase[(size_t)(p % Wn) * 2 * HD + HD + d] = v16;
        Vs[j * LD + d] = bf2f(v16);
    }
    __syncthreads();
    const float scale = rsqrtf((float)D);
    for (int pr = tid; pr < T * Wn; pr += 256) {
        const int i = pr / Wn, j = pr % Wn;
        const int nvis = Wn - T + i + 1;
        float sc = -INFINITY;
        if (j < nvis) {
            const float* q = Qs + (size_t)i * D;
            const float* kr = Ks + (size_t)j * LD;
            sc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) sc = fmaf(q[d], kr[d], sc);
            sc *= scale;
        }
        Ps[pr] = sc;
    }
    __syncthreads();
    for (int i = wave; i < T; i += 4) {
        float* P = Ps + (size_t)i * Wn;
        const int nvis = Wn - T + i + 1;
        float mx = -INFINITY;
        for (int j = lane; j < Wn; j += 64) mx = fmaxf(mx, P[j]);
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float ls = 0.0f;
        for (int j = lane; j < Wn; j += 64) {
            const float pj = j < nvis ? expf(P[j] - mx) : 0.0f;
            P[j] = pj;
            ls += pj;
        }
        for (int off = 32; off >= 1; off >>= 1) ls += __shfl_xor(ls, off, 64);
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (int d = lane; d < D; d += 64) {
            float o = 0.0f;
            int j = 0;
            for (; j + 8 <= nvis; j += 8) {
                float pv[8], vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { pv[u] = P[j + u]; vv[u] = Vs[(j + u) * LD + d]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) o = fmaf(pv[u], vv[u], o);
            }
            for (; j < nvis; ++j) o = fmaf(P[j], Vs[j * LD + d], o);
            out[((size_t)(b * T + i)) * HD + (size_t)h * D + d] = o / ls;
        }
    }
}

__global__ __launch_bounds__(256) void k_pos_advance(long* pos, const int* slots, int n, int T) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < n) pos[slots[b]] += T;
}

// h[m][i] = silu(gu[m][i]) * gu[m][I+i]
__global__ __launch_bounds__(256) void k_silu_mul_f32(const float* gu, float* h, size_t rows, int I) {
    const size_t total = rows * I;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t m = e / I;
        const int i = (int)(e % I);
        const float g = gu[m * 2 * I + i], u = gu[m * 2 * I + I + i];
        h[e] = (g / (1.0f + expf(-g))) * u;
    }
}

// ConvNeXt front: causal depthwise conv k=7 (+bias) then LayerNorm over channels.   qwen3_codec.py:447-450
__global__ __launch_bounds__(256) void k_dwconv_ln(const float* x, const float* state, const int* slots, const float* w,
                                                   const float* wb, const float* lnw, const float* lnb, float* y, int L,
                                                   int C, float eps) {
    __shared__ float red[4];
    extern __shared__ float ybuf[];   // [C]
    const int row = blockIdx.x, b = row / L, t = row % L;
    const float* st = state + (size_t)slots[b] * 6 * C;
    float s1 = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = wb[c];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int e = t + k;   // index into ext = [state(6) ++ x]
            const float v = e < 6 ? st[(size_t)e * C + c] : x[((size_t)b * L + (e - 6)) * C + c];
            acc += w[c * 7 + k] * v;
        }
        ybuf[c] = acc;
        s1 += acc;
    }
    const float mean = block_sum(s1, red) / (float)C;
    float s2 = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float d = ybuf[c] - mean;
        s2 += d * d;
    }
    const float rstd = rsqrtf(block_sum(s2, red) / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) y[(size_t)row * C + c] = (ybuf[c] - mean) * rstd * lnw[c] + lnb[c];
}

// final causal conv k=7, C -> 1, + clamp(-1,1).  One wave per output sample group.
__global__ __launch_bounds__(256) void k_final_conv(const float* x, const float* state, const int* slots, const float* w,
                                                    float bias, float* out, int L, int C) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= L) return;
    const float* st = state + (size_t)slots[b] * 6 * C;
    float acc = 0.0f;
    for (int e = lane; e < 7 * C; e += 64) {
        const int k = e / C, c = e % C;
        const int r = t + k;
        const float v = r < 6 ? st[(size_t)r * C + c] : x[((size_t)b * L + (r - 6)) * C + c];
        acc += w[c * 7 + k] * v;
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[(size_t)b * L + t] = fminf(1.0f, fmaxf(-1.0f, acc + bias));
}

// The same conv for many rows: 128 consecutive outputs of one request per block (2 waves, one output per lane).  The 134 input rows the
// block needs are staged in LDS once (coalesced 16-byte loads; row stride C + 1 words, so the 64 lanes of a wave — 64 consecutive rows,
// same channel — hit 64 different banks), the weights beside them (read as broadcasts).  The one-wave-per-output form above re-reads
// every input row 7 times through 4-byte strided loads with an integer division per element: 518 us for 614 k outputs (236 MB of input).
// Per output: four partial sums over the channel residues mod 4, taps outermost, added ((s0 + s1) + (s2 + s3)) + bias.
__global__ __launch_bounds__(128) void k_final_conv_rows(const float* x, const float* state, const int* slots, const float* w,
                                                         float bias, float* out, int L, int C) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];      // [134][C + 1] rows, then [7][C] weights (tap-major)
    const int b = blockIdx.y, t0 = blockIdx.x * 128, LD = C + 1;
    float* ws = fsm + 134 * LD;
    const float* st = state + (size_t)slots[b] * 6 * C;
    const float* xb = x + (size_t)b * L * C;
    const int c4n = C >> 2, total = 134 * c4n;
    for (int e0 = 0; e0 < total; e0 += 128 * 8) {      // eight requests per thread in flight, then their LDS stores
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = e0 + i * 128 + threadIdx.x, j = e / c4n, c4 = e - j * c4n, r = t0 + j;      // r indexes [state(6) ++ x]
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < total) {
                if (r < 6) v[i] = *reinterpret_cast<const float4*>(st + (size_t)r * C + c4 * 4);
                else if (r - 6 < L) v[i] = *reinterpret_cast<const float4*>(xb + (size_t)(r - 6) * C + c4 * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = e0 + i * 128 + threadIdx.x, j = e / c4n, c4 = e - j * c4n;
            if (e < total) {
                float* d = fsm + j * LD + c4 * 4;
                d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
            }
        }
    }
    for (int e = threadIdx.x; e < 7 * C; e += 128) { const int k = e / C, c = e - k * C; ws[e] = w[c * 7 + k]; }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t >= L) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k = 0; k < 7; ++k) {
        const float* xr = fsm + (threadIdx.x + k) * LD;
        const float* wk = ws + k * C;
        for (int c = 0; c < C; c += 4) {
            s0 = __fmaf_rn(wk[c], xr[c], s0); s1 = __fmaf_rn(wk[c + 1], xr[c + 1], s1);
            s2 = __fmaf_rn(wk[c + 2], xr[c + 2], s2); s3 = __fmaf_rn(wk[c + 3], xr[c + 3], s3);
        }
    }
    out[(size_t)b * L + t] = fminf(1.0f, fmaxf(-1.0f, ((s0 + s1) + (s2 + s3)) + bias));
}

// ================================================================================================
// engine
// ================================================================================================
#include <vector>

struct vox_codec {
    vox_ctx* ctx;
    vox_codec_config cfg;
    vox_codec_weights w;
    int max_batch, max_slots, T;
    int planes = 2;     // operand planes of the conv GEMMs (vox_codec_set_operand_planes)
    bool decoded = false;   // a chunk has run: the slots' streaming history is stored in the format `planes` implies
    // state (per slot)
    float *st_pre, *st_dw[2], *st_dec0, *st_tc[4], *st_ru[4][3], *st_final;
    bf16_t* ring;   // [layers][slots][Wn][2][HD]
    long* pos;      // [slots]
    // work buffers
    float* buf[4];
    size_t buf_floats;
    float *q0, *qr;
};

// operand planes of the conv GEMMs launched by the current decode call (set by the entry points from their object's setting)
static thread_local int g_conv_planes = 3;
// rows up to which the few-row GEMM variant (16-row tiles, wide K steps) is used: the flow's GEMMs have 56..450 rows and K <= 2048, their
// time is the number of dependent K steps, not MFMA issue
static thread_local int g_conv_skinny_rows = 48;
// k_rows_gemm for eligible shapes: the flows only (their row bound is unbounded, so one kernel serves a shape at every batch size; the
// codecs' and HiFT's few-row stages become many-row stages as the batch grows and must keep the staged kernels' summation order)
static thread_local bool g_conv_rows_gemm = false;
// few-row GEMMs walk K in 256-wide steps where Cin allows (half the dependent steps of the 128-wide walk; same accumulation order, so
// the results are bit-identical).  VOX_SKINNY_BK256=0 keeps the 128-wide walk (A/B timing).
static bool skinny_wide() {
    static const bool on = [] { const char* e = getenv("VOX_SKINNY_BK256"); return !(e && e[0] == '0'); }();
    return on;
}
// VOX_ROWS_GEMM=0: the LDS-staged few-row kernel instead of k_rows_gemm (A/B timing)
static bool rows_gemm_on() {
    static const bool on = [] { const char* e = getenv("VOX_ROWS_GEMM"); return !(e && e[0] == '0'); }();
    return on;
}
// VOX_ROWS_WV8=0: four waves per k_rows_gemm block at every K (A/B timing; changes the summation order of the K >= 512 linears)
static bool rows_wv8_on() {
    static const bool on = [] { const char* e = getenv("VOX_ROWS_WV8"); return !(e && e[0] == '0'); }();
    return on;
}
// blocks from which k_rows_gemm takes two 16-row tiles per block (VOX_ROWS_MT2=<blocks>; 0 = never)
static int rows_mt2_blocks() {
    static const int v = [] { const char* e = getenv("VOX_ROWS_MT2"); const int x = e ? atoi(e) : 1024; return x > 0 ? x : (1 << 30); }();
    return v;
}
// 16 x 16 output tiles above which a k_rows_gemm block takes 2 / 4 column tiles (VOX_ROWS_NT2 / VOX_ROWS_NT4)
static int rows_nt_tiles(int which) {
    static const int v2 = [] { const char* e = getenv("VOX_ROWS_NT2"); return e ? atoi(e) : 256; }();
    static const int v4 = [] { const char* e = getenv("VOX_ROWS_NT4"); return e ? atoi(e) : 2048; }();
    return which ? v4 : v2;
}
// VOX_CONV_TAPS=0: the per-tap staging kernel for every multi-tap conv (A/B timing)
static bool conv_taps_on() {
    static const bool on = [] { const char* e = getenv("VOX_CONV_TAPS"); return !(e && e[0] == '0'); }();
    return on;
}
static int conv_gemm(hipStream_t st, const vox_conv_w& w, const float* x, const float* state, const int* slots, int n,
                     int L, int P, const int* offs, float* out, const float* res, const float* scale, int gelu,
                     float* out2 = nullptr, const vox_snake_w* sn = nullptr, int sn_mod = 0, const float* rscale = nullptr,
                     const float* ln_w = nullptr, const float* ln_b = nullptr, float ln_eps = 0.0f, int x_s2 = 0, int out2_s2 = 0) {
    if (w.cin % CG_BK) return vox_fail(VOX_ERR_INVALID, "codec gemm: Cin %d %% %d != 0", w.cin, CG_BK);
    if (w.n_taps > CG_MAXTAPS) return vox_fail(VOX_ERR_INVALID, "codec gemm: too many taps");
    ConvGemmArgs a{};
    a.x = x; a.state = state; a.slots = slots; a.w = (const bf16_t*)w.w; a.bias = w.bias; a.res = res; a.scale = scale;
    a.out = out; a.M = n * L; a.N = w.n; a.Cin = w.cin; a.L = L; a.P = P; a.n_taps = w.n_taps; a.rscale = rscale;
    a.planes = x_s2 ? 2 : g_conv_planes;
    a.x_s2 = x_s2; a.out2_s2 = (out2 && sn) ? out2_s2 : 0;
#ifdef VOX_DEV_KNOBS
    { static const int dv = [] { const char* e = getenv("VOX_CODEC_DEV"); return e ? atoi(e) : 0; }(); a.dev = dv; }
#endif
    if (x_s2 && ln_w) return vox_fail(VOX_ERR_INVALID, "codec gemm: S2 rows with a fused LayerNorm");
    if (a.out2_s2 && ((w.n & 1) || ((sn_mod > 0 ? sn_mod : w.n) & 1))) return vox_fail(VOX_ERR_INVALID, "codec gemm: S2 output needs an even channel count");
    a.bias_mod = w.bias_mod > 0 ? w.bias_mod : w.n; a.gelu = gelu;
    if (out2 && sn) { a.out2 = out2; a.sn_alpha = sn->alpha; a.sn_invb = sn->inv_beta; a.sn_mod = sn_mod > 0 ? sn_mod : w.n; }
    for (int k = 0; k < w.n_taps; ++k) a.off[k] = offs ? offs[k] : 0;
    if (g_conv_rows_gemm && rows_gemm_on() && !x_s2 && a.M <= g_conv_skinny_rows && w.cin % 32 == 0 && (w.n_taps * w.cin) % 128 == 0 && (!ln_w || (w.n_taps == 1 && a.off[0] == 0))) {
        // every operand requested up front, K split over the block's waves (k_rows_gemm): 8 waves from K = 512 on
        const int K = w.n_taps * w.cin, wv = (K >= 512 && K % 256 == 0 && rows_wv8_on()) ? 8 : 4, kb = K / (32 * wv);
        const int tiles = ((w.n + 15) / 16) * ((a.M + 15) / 16);
        // column tiles per block: one while the call is a handful of row tiles (latency: more blocks in flight), two from 256 tiles on
        // once there are >= 16 row tiles (every block re-loads and re-splits its A rows: at 448+ rows that traffic and VALU work
        // outweigh the extra blocks — CosyVoice2 chunk 49.7 -> 46.6 ms at B=8, 36.6 -> 35.2 at B=4, unchanged at B=1), four above 2048
        int nt = tiles > rows_nt_tiles(1) ? 4 : tiles > (a.M >= 256 ? rows_nt_tiles(0) : 1024) ? 2 : 1;
        while (nt > 1 && nt * kb > 16) nt >>= 1;
        const int mt = (tiles / nt >= rows_mt2_blocks() && kb <= 4 && (wv == 4 || nt <= 2)) ? 2 : 1;      // (the 8-wave sum buffer: MT NT <= 4)
        if (ln_w) { a.ln_w = ln_w; a.ln_b = ln_b; a.ln_eps = ln_eps; }
        const dim3 g((w.n + 16 * nt - 1) / (16 * nt), (a.M + 16 * mt - 1) / (16 * mt));
#define VOX_ROWS1(KB_, NT_, MT_, WV_)                                                                              \
        if (kb == KB_ && nt == NT_ && mt == MT_ && wv == WV_) {                                                        \
            if (ln_w) hipLaunchKernelGGL((k_rows_gemm<KB_, NT_, true, MT_, WV_>), g, dim3(64 * WV_), 0, st, a);        \
            else hipLaunchKernelGGL((k_rows_gemm<KB_, NT_, false, MT_, WV_>), g, dim3(64 * WV_), 0, st, a);            \
            return VOX_OK;                                                                                             \
        }
#define VOX_ROWS(KB_, WV_) VOX_ROWS1(KB_, 1, 1, WV_) VOX_ROWS1(KB_, 2, 1, WV_) VOX_ROWS1(KB_, 4, 1, WV_) VOX_ROWS1(KB_, 1, 2, WV_) \
                           VOX_ROWS1(KB_, 2, 2, WV_)
        VOX_ROWS(2, 4) VOX_ROWS(4, 4) VOX_ROWS(2, 8) VOX_ROWS(3, 8) VOX_ROWS(4, 8) VOX_ROWS1(2, 4, 2, 4) VOX_ROWS1(4, 4, 2, 4)
        VOX_ROWS1(6, 1, 1, 4) VOX_ROWS1(6, 2, 1, 4) VOX_ROWS1(8, 1, 1, 4) VOX_ROWS1(8, 2, 1, 4) VOX_ROWS1(12, 1, 1, 4) VOX_ROWS1(16, 1, 1, 4)
        VOX_ROWS1(6, 1, 1, 8) VOX_ROWS1(6, 2, 1, 8) VOX_ROWS1(8, 1, 1, 8) VOX_ROWS1(8, 2, 1, 8)
#undef VOX_ROWS
#undef VOX_ROWS1
        a.ln_w = a.ln_b = nullptr;      // no variant for this K: the staged kernels below
    }
    if (ln_w) {      // fused input LayerNorm: few-row kernel, one plain tap, the row statistics need the whole row in one K walk
        if (a.M > g_conv_skinny_rows || w.n_taps != 1 || a.off[0] != 0)
            return vox_fail(VOX_ERR_INVALID, "codec gemm: fused LayerNorm needs the few-row one-tap path");
        a.ln_w = ln_w; a.ln_b = ln_b; a.ln_eps = ln_eps;
        const dim3 g((w.n + 63) / 64, (a.M + 15) / 16);
        if (w.cin % 256 == 0 && skinny_wide()) hipLaunchKernelGGL((k_conv_gemm_skinny<256, true>), g, dim3(256), 0, st, a);
        else if (w.cin % 128 == 0) hipLaunchKernelGGL((k_conv_gemm_skinny<128, true>), g, dim3(256), 0, st, a);
        else if (w.cin % 64 == 0) hipLaunchKernelGGL((k_conv_gemm_skinny<64, true>), g, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_conv_gemm_skinny<32, true>), g, dim3(256), 0, st, a);
        return VOX_OK;
    }
    if (a.M <= g_conv_skinny_rows) {
        const dim3 g((w.n + 63) / 64, (a.M + 15) / 16);
        if (w.cin % 256 == 0 && skinny_wide()) hipLaunchKernelGGL(k_conv_gemm_skinny<256>, g, dim3(256), 0, st, a);
        else if (w.cin % 128 == 0) hipLaunchKernelGGL(k_conv_gemm_skinny<128>, g, dim3(256), 0, st, a);
        else if (w.cin % 64 == 0) hipLaunchKernelGGL(k_conv_gemm_skinny<64>, g, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_conv_gemm_skinny<32>, g, dim3(256), 0, st, a);
        return VOX_OK;
    }
    // many-row stages (batched chunks, the late upsampling stages): 128-row tiles, 64 x (48|64) per wave — each LDS
    // fragment read then feeds 3-4 MFMAs instead of 2 (the 64 x 64 tile is LDS-read bound with the 3-term split)
    {
        const int wnb = (w.n % 128 == 0 || w.n >= 256) ? 4 : ((w.n > 64 && w.n <= 96) || w.n % 96 == 0) ? 3 : 0;
        // S2 rows and several taps: the activation tile is staged once per 32-channel chunk for all taps (k_conv_taps); 128-row tiles when
        // they fill the chip and divide the request length, else 64-row tiles
        if (wnb && x_s2 && w.n_taps > 1 && conv_taps_on()) {
            int H = 0, omin = 0;
            for (int k = 0; k < w.n_taps; ++k) { H = a.off[k] > H ? a.off[k] : H; omin = a.off[k] < omin ? a.off[k] : omin; }
            const int ncb = (w.n + 32 * wnb - 1) / (32 * wnb);
            // (below ~256 blocks the per-tap kernels' smaller tiles win: measured at 1 and 4 requests)
            const int bm = (L % 128 == 0 && ncb * (a.M / 128) >= 256) ? 128 : ((L % 64 == 0 && ncb * (a.M / 64) >= 256) ? 64 : 0);
            if (bm && omin >= 0 && H <= 64) {
                const dim3 gt(ncb, a.M / bm);
                const size_t lds = (size_t)(2 * (bm + H) + 2 * 32 * wnb) * 80;
                if (bm == 128 && wnb == 4) hipLaunchKernelGGL((k_conv_taps<4, 4>), gt, dim3(256), lds, st, a, H);
                else if (bm == 128) hipLaunchKernelGGL((k_conv_taps<4, 3>), gt, dim3(256), lds, st, a, H);
                else if (wnb == 4) hipLaunchKernelGGL((k_conv_taps<2, 4>), gt, dim3(256), lds, st, a, H);
                else hipLaunchKernelGGL((k_conv_taps<2, 3>), gt, dim3(256), lds, st, a, H);
                return VOX_OK;
            }
        }
        if (wnb && ((w.n + 32 * wnb - 1) / (32 * wnb)) * ((a.M + 127) / 128) >= 256) {
            const dim3 gb((w.n + 32 * wnb - 1) / (32 * wnb), (a.M + 127) / 128);
            if (wnb == 4) hipLaunchKernelGGL((k_conv_gemm<32, 4, 4>), gb, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_conv_gemm<32, 4, 3>), gb, dim3(256), 0, st, a);
            return VOX_OK;
        }
    }
    // tile: 64 x 64 unless that leaves most CUs idle (these GEMMs are MFMA-bound per CU)
    const int b64 = ((w.n + 63) / 64) * ((a.M + 63) / 64);
    const int wm = b64 >= 192 ? 2 : 1, wn = (b64 >= 192 || 2 * b64 >= 192) ? 2 : 1;
    const dim3 grid((w.n + 32 * wn - 1) / (32 * wn), (a.M + 32 * wm - 1) / (32 * wm));
    const bool few = grid.x * grid.y <= 512;       // at most two blocks per CU: nothing else hides a block's LDS round trips
#define VOX_CG(K_, M_, N_) if (wm == M_ && wn == N_) { \
        if (few && M_ * N_ <= 2) hipLaunchKernelGGL((k_conv_gemm<K_, M_, N_, (M_ * N_ <= 2)>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((k_conv_gemm<K_, M_, N_>), grid, dim3(256), 0, st, a); \
        return VOX_OK; }
    // few blocks, long K (HiFT's resblock convs of one request: 112 blocks x 88 K steps of 64): 128-wide K steps halve the dependent
    // stage-and-barrier rounds; the k order is unchanged, so the results are bit-identical (VOX_CG_BK128=0: 64-wide steps)
    static const bool bk128 = [] { const char* e = getenv("VOX_CG_BK128"); return !(e && e[0] == '0'); }();
    if (w.cin % 128 == 0 && bk128) { VOX_CG(128, 1, 1) }
    if (w.cin % 64 == 0) { VOX_CG(64, 2, 2) VOX_CG(64, 1, 2) VOX_CG(64, 1, 1) }
    VOX_CG(32, 2, 2) VOX_CG(32, 1, 2) VOX_CG(32, 1, 1)
#undef VOX_CG
    return vox_fail(VOX_ERR_INVALID, "codec gemm: no tile variant");
}

// A residual unit as ONE launch (k_res_unit) where the tile can cover all channels and the stage has enough rows to fill the chip;
// *fused = 0 and nothing launched otherwise (the caller then runs the two convs).  x: the S2 activated input rows; h: the fp32
// residual stream, updated in place; out2: the NEXT consumer's activation of the new h — must not alias x.
static bool res_unit_on() {
    static const bool on = [] { const char* e = getenv("VOX_RES_UNIT"); return !(e && e[0] == '0'); }();
    return on;
}
// (measured at 614 k / 205 k rows, B = 32: the 96-channel unit 544 -> 390-430 us fused; the 192-channel unit with the 64 x 192 tile it
// needs 482 -> 500 us — its two launches stay.  VOX_RES_UNIT_192=1 fuses it all the same, for timing.)
static bool res_unit_192() {
    static const bool on = [] { const char* e = getenv("VOX_RES_UNIT_192"); return e && e[0] == '1'; }();
    return on;
}
static int conv_unit(hipStream_t st, const vox_conv_w& w1, const vox_conv_w& w2, const float* x, const float* state, const int* slots, int n, int L,
                     int P, const int* offs, const vox_snake_w& act2, float* h, float* out2, const vox_snake_w* nx, int out2_s2, int* fused) {
    *fused = 0;
    const int C = w1.n;
    if (!res_unit_on() || !conv_taps_on() || g_conv_planes != 2 || w1.n_taps < 2 || w1.n_taps > CG_MAXTAPS || w2.n_taps != 1 || w1.cin != C ||
        w2.cin != C || w2.n != C || (C != 96 && !(C == 192 && res_unit_192())) || out2 == x || !nx)
        return VOX_OK;
    int H = 0, omin = 0;
    for (int k = 0; k < w1.n_taps; ++k) { H = offs[k] > H ? offs[k] : H; omin = offs[k] < omin ? offs[k] : omin; }
    const int M = n * L, bm = C == 96 ? 128 : 64;
    if (omin < 0 || H > 64 || L % bm || M / bm < 256) return VOX_OK;
    ConvGemmArgs a1{}, a2{};
    a1.x = x; a1.state = state; a1.slots = slots; a1.w = (const bf16_t*)w1.w; a1.bias = w1.bias; a1.M = M; a1.N = C; a1.Cin = C; a1.L = L; a1.P = P;
    a1.n_taps = w1.n_taps; a1.planes = 2; a1.x_s2 = 1; a1.bias_mod = w1.bias_mod > 0 ? w1.bias_mod : C;
    a1.sn_alpha = act2.alpha; a1.sn_invb = act2.inv_beta; a1.sn_mod = C;
    for (int k = 0; k < w1.n_taps; ++k) a1.off[k] = offs[k];
    a2.w = (const bf16_t*)w2.w; a2.bias = w2.bias; a2.res = h; a2.out = h; a2.M = M; a2.N = C; a2.Cin = C; a2.L = L; a2.n_taps = 1; a2.planes = 2;
    a2.x_s2 = 1; a2.bias_mod = w2.bias_mod > 0 ? w2.bias_mod : C; a2.out2 = out2; a2.sn_alpha = nx->alpha; a2.sn_invb = nx->inv_beta; a2.sn_mod = C;
    a2.out2_s2 = out2_s2;
    const size_t p1 = (size_t)(2 * (bm + H) + 2 * C) * 80, p3 = (size_t)2 * bm * (C + 8) * 2 + (size_t)2 * C * 80;
    const size_t lds = p1 > p3 ? p1 : p3;
    const dim3 g(1, M / bm);
    // (the > 64 KB dynamic-LDS opt-in is a per-device function attribute: the detokenizer may live on a second GPU)
    static bool attr_done[2][16] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& done = attr_done[C == 96 ? 0 : 1][dev_id & 15];
    if (C == 96) {
        if (!done && lds > 64 * 1024) { VOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_res_unit<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL((k_res_unit<4, 3>), g, dim3(256), lds, st, a1, a2, H);
    } else {
        if (!done && lds > 64 * 1024) { VOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_res_unit<2, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL((k_res_unit<2, 6>), g, dim3(256), lds, st, a1, a2, H);
    }
    *fused = 1;
    return VOX_OK;
}

static void snake(hipStream_t st, const float* x, const vox_snake_w& s, float* y, size_t rows, int C) {
    const size_t total = rows * C;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_snake, dim3(grid), dim3(256), 0, st, x, s.alpha, s.inv_beta, y, total, C);
}
static void state_update(hipStream_t st, float* state, const int* slots, const float* x, int n, int L, int P, int C) {
    hipLaunchKernelGGL(k_state_update, dim3((C + 255) / 256, n), dim3(256), 0, st, state, slots, x, L, P, C);
}

extern "C" {

int vox_codec_create(vox_ctx* ctx, const vox_codec_config* cfg, const vox_codec_weights* w, int max_batch, int max_slots,
                     int frames_per_chunk, vox_codec** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "codec_create: NULL");
    if (cfg->n_blocks != 4 || cfg->n_upsample != 2 || cfg->num_layers > 16)
        return vox_fail(VOX_ERR_INVALID, "codec_create: expects 4 decoder blocks, 2 upsample stages");
    vox_codec* m = new vox_codec();
    m->ctx = ctx; m->cfg = *cfg; m->w = *w; m->max_batch = max_batch; m->max_slots = max_slots; m->T = frames_per_chunk;
    const int S = max_slots, HD = cfg->num_heads * cfg->head_dim;
    size_t L = frames_per_chunk;
    auto zalloc = [&](float** p, size_t n) { return hipMalloc((void**)p, n * 4) == hipSuccess && hipMemset(*p, 0, n * 4) == hipSuccess; };
    bool ok = zalloc(&m->st_pre, (size_t)S * 2 * cfg->codebook_dim);
    for (int u = 0; u < 2; ++u) ok = ok && zalloc(&m->st_dw[u], (size_t)S * 6 * cfg->latent_dim);
    ok = ok && zalloc(&m->st_dec0, (size_t)S * 6 * cfg->latent_dim);
    size_t maxf = (size_t)L * (4 * cfg->latent_dim) * 4;   // ConvNeXt hidden at 4L rows
    L *= 4;
    for (int b = 0; b < 4; ++b) {
        const int cin = cfg->decoder_dim >> b, cout = cfg->decoder_dim >> (b + 1);
        ok = ok && zalloc(&m->st_tc[b], (size_t)S * cin);
        L *= cfg->rates[b];
        for (int u = 0; u < 3; ++u) {
            const int d = u == 0 ? 1 : (u == 1 ? 3 : 9);
            ok = ok && zalloc(&m->st_ru[b][u], (size_t)S * 6 * d * cout);
        }
        if (L * cout > maxf) maxf = L * cout;
        if ((L / cfg->rates[b]) * cin > maxf) maxf = (L / cfg->rates[b]) * cin;
    }
    ok = ok && zalloc(&m->st_final, (size_t)S * 6 * (cfg->decoder_dim >> 4));
    const size_t ring_elems = (size_t)cfg->num_layers * S * cfg->window * 2 * HD;
    ok = ok && hipMalloc((void**)&m->ring, ring_elems * 2) == hipSuccess && hipMemset(m->ring, 0, ring_elems * 2) == hipSuccess;
    ok = ok && hipMalloc((void**)&m->pos, (size_t)S * 8) == hipSuccess && hipMemset(m->pos, 0, (size_t)S * 8) == hipSuccess;
    m->buf_floats = maxf * max_batch;
    for (int i = 0; i < 4; ++i) ok = ok && hipMalloc((void**)&m->buf[i], m->buf_floats * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&m->q0, (size_t)max_batch * frames_per_chunk * cfg->vq_dim * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&m->qr, (size_t)max_batch * frames_per_chunk * cfg->vq_dim * 4) == hipSuccess;
    if (!ok) return vox_fail(VOX_ERR_NOMEM, "codec_create: hipMalloc failed");
    *out = m;
    return VOX_OK;
}

int vox_codec_set_operand_planes(vox_codec* m, int planes) {
    if (!m || planes < 1 || planes > 3) return vox_fail(VOX_ERR_INVALID, "codec_set_operand_planes: 1, 2 or 3");
    // the slots' history rows are kept in the operand format (two-term rows vs fp32 rows): switching with history in place would
    // make the convs misread it, so the mode is fixed once the first chunk has been decoded
    if (planes != m->planes && m->decoded)
        return vox_fail(VOX_ERR_INVALID, "codec_set_operand_planes: %d -> %d after a chunk was decoded (set it before the first chunk)", m->planes, planes);
    m->planes = planes;
    return VOX_OK;
}
void vox_codec_destroy(vox_codec* m) {
    if (!m) return;
    (void)hipFree(m->st_pre); (void)hipFree(m->st_dec0); (void)hipFree(m->st_final); (void)hipFree(m->ring);
    (void)hipFree(m->pos); (void)hipFree(m->q0); (void)hipFree(m->qr);
    for (int u = 0; u < 2; ++u) (void)hipFree(m->st_dw[u]);
    for (int b = 0; b < 4; ++b) {
        (void)hipFree(m->st_tc[b]);
        for (int u = 0; u < 3; ++u) (void)hipFree(m->st_ru[b][u]);
    }
    for (int i = 0; i < 4; ++i) (void)hipFree(m->buf[i]);
    delete m;
}

int64_t vox_codec_state_bytes(vox_codec* m) {
    if (!m) return 0;
    const vox_codec_config& c = m->cfg;
    int64_t f = 2 * c.codebook_dim + 6 * c.latent_dim * 3 + 6 * (c.decoder_dim >> 4);
    for (int b = 0; b < 4; ++b) f += (c.decoder_dim >> b) + 6 * 13 * (c.decoder_dim >> (b + 1));
    return f * 4 + (int64_t)c.num_layers * c.window * 2 * c.num_heads * c.head_dim * 2 + 8;
}

// Zero the streaming state of one slot (a new request takes it over).  Replaces audio_decoder_initial_cache
// (model/qwen3_tts.py:1243-1260).
int vox_codec_reset_slot(vox_codec* m, void* stream, int slot) {
    if (!m || slot < 0 || slot >= m->max_slots) return vox_fail(VOX_ERR_INVALID, "codec_reset_slot: bad slot");
    hipStream_t st = (hipStream_t)stream;
    const vox_codec_config& c = m->cfg;
    auto z = [&](float* p, size_t per) { return hipMemsetAsync(p + (size_t)slot * per, 0, per * 4, st); };
    VOX_HIP(z(m->st_pre, (size_t)2 * c.codebook_dim));
    for (int u = 0; u < 2; ++u) VOX_HIP(z(m->st_dw[u], (size_t)6 * c.latent_dim));
    VOX_HIP(z(m->st_dec0, (size_t)6 * c.latent_dim));
    for (int b = 0; b < 4; ++b) {
        VOX_HIP(z(m->st_tc[b], (size_t)(c.decoder_dim >> b)));
        for (int u = 0; u < 3; ++u) VOX_HIP(z(m->st_ru[b][u], (size_t)6 * (u == 0 ? 1 : (u == 1 ? 3 : 9)) * (c.decoder_dim >> (b + 1))));
    }
    VOX_HIP(z(m->st_final, (size_t)6 * (c.decoder_dim >> 4)));
    const size_t HD = (size_t)c.num_heads * c.head_dim, per = (size_t)c.window * 2 * HD;
    for (int l = 0; l < c.num_layers; ++l)
        VOX_HIP(hipMemsetAsync(m->ring + ((size_t)l * m->max_slots + slot) * per, 0, per * 2, st));
    VOX_HIP(hipMemsetAsync(m->pos + slot, 0, 8, st));
    return VOX_OK;
}

// codes: int32 [n, T, code_stride] (first num_quantizers columns used); slots: int32 [n]; out: fp32 [n, T*hop]
int vox_codec_decode_chunk(vox_codec* m, void* stream, const int32_t* codes, int code_stride, const int32_t* slots, int n,
                           int T, float* out) {
    if (!m || !codes || !slots || !out) return vox_fail(VOX_ERR_INVALID, "codec_decode_chunk: NULL");
    if (n < 1 || n > m->max_batch || T < 1 || T > m->T) return vox_fail(VOX_ERR_INVALID, "codec_decode_chunk: n=%d T=%d out of range", n, T);
    m->decoded = true;
    hipStream_t st = (hipStream_t)stream;
    struct PlanesGuard { int saved; explicit PlanesGuard(int p) : saved(g_conv_planes) { g_conv_planes = p; } ~PlanesGuard() { g_conv_planes = saved; } } pg(m->planes);
    const vox_codec_config& c = m->cfg;
    const vox_codec_weights& w = m->w;
    const int HD = c.num_heads * c.head_dim, H = c.hidden, LD = c.latent_dim;
    float *A = m->buf[0], *B = m->buf[1], *C = m->buf[2], *D = m->buf[3];
    const int off0[1] = {0};
    int L = T;

    hipLaunchKernelGGL(k_rvq, dim3(n * L), dim3(256), 0, st, codes, code_stride, w.emb, c.num_quantizers, c.codebook_size,
                       c.vq_dim, m->q0, m->qr);
    VOX_TRY(conv_gemm(st, w.rvq_first_out, m->q0, nullptr, slots, n, L, 0, off0, A, nullptr, nullptr, 0));
    VOX_TRY(conv_gemm(st, w.rvq_rest_out, m->qr, nullptr, slots, n, L, 0, off0, B, A, nullptr, 0));      // B = A + rest
    const int off3[3] = {2, 1, 0};
    VOX_TRY(conv_gemm(st, w.pre_conv, B, m->st_pre, slots, n, L, 2, off3, A, nullptr, nullptr, 0));      // [nL, latent]
    state_update(st, m->st_pre, slots, B, n, L, 2, c.codebook_dim);
    // ---- transformer (hidden H) ----
    VOX_TRY(conv_gemm(st, w.in_proj, A, nullptr, slots, n, L, 0, off0, B, nullptr, nullptr, 0));          // h = B [nL,H]
    for (int l = 0; l < c.num_layers; ++l) {
        const vox_codec_layer_w& lw = w.layers[l];
        bf16_t* ring = m->ring + (size_t)l * m->max_slots * c.window * 2 * HD;
        hipLaunchKernelGGL(k_rmsnorm_f32, dim3(n * L), dim3(256), 0, st, B, lw.ln1, C, H, c.rms_eps);
        VOX_TRY(conv_gemm(st, lw.qkv, C, nullptr, slots, n, L, 0, off0, D, nullptr, nullptr, 0));        // D [nL,3HD]
        // (VOX_CODEC_ATTN2=0: RoPE / ring write and the attention as two launches — A/B timing; bit-identical)
        static const bool attn2 = [] { const char* e = getenv("VOX_CODEC_ATTN2"); return !(e && e[0] == '0'); }();
        const size_t lds2 = (size_t)(2 * c.window * (c.head_dim + 1) + L * c.head_dim + L * c.window) * 4;
        if (attn2 && c.head_dim == 64 && L <= c.window && lds2 <= 64 * 1024) {
            hipLaunchKernelGGL(k_codec_attn2<64>, dim3(c.num_heads, n), dim3(256), lds2,
                               st, D, ring, slots, m->pos, C, L, c.num_heads, c.window, w.inv_freq);       // C [nL,HD]
        } else {
            hipLaunchKernelGGL(k_codec_rope_kv, dim3(n * L), dim3(256), 0, st, D, ring, slots, m->pos, L, c.num_heads,
                               c.head_dim, c.window, w.inv_freq);
            hipLaunchKernelGGL(k_codec_attn, dim3(c.num_heads, n), dim3(256),
                               (size_t)(2 * c.window * (c.head_dim + 1) + L * c.head_dim + 4 * c.window) * 4,
                               st, D, ring, slots, m->pos, C, L, c.num_heads, c.head_dim, c.window);         // C [nL,HD]
        }
        VOX_TRY(conv_gemm(st, lw.o, C, nullptr, slots, n, L, 0, off0, B, B, lw.scale1, 0));              // h += s1*o(attn)
        hipLaunchKernelGGL(k_rmsnorm_f32, dim3(n * L), dim3(256), 0, st, B, lw.ln2, C, H, c.rms_eps);
        VOX_TRY(conv_gemm(st, lw.gate_up, C, nullptr, slots, n, L, 0, off0, D, nullptr, nullptr, 0));    // D [nL,2I]
        hipLaunchKernelGGL(k_silu_mul_f32, dim3((n * L * c.intermediate + 255) / 256), dim3(256), 0, st, D, C,
                           (size_t)n * L, c.intermediate);
        VOX_TRY(conv_gemm(st, lw.down, C, nullptr, slots, n, L, 0, off0, B, B, lw.scale2, 0));           // h += s2*down
    }
    hipLaunchKernelGGL(k_pos_advance, dim3((n + 255) / 256), dim3(256), 0, st, m->pos, slots, n, L);
    hipLaunchKernelGGL(k_rmsnorm_f32, dim3(n * L), dim3(256), 0, st, B, w.final_norm, C, H, c.rms_eps);
    VOX_TRY(conv_gemm(st, w.out_proj, C, nullptr, slots, n, L, 0, off0, A, nullptr, nullptr, 0));         // A [nL,latent]
    // ---- upsample x2 x2 with ConvNeXt ----
    for (int u = 0; u < 2; ++u) {
        const vox_codec_up_w& uw = w.up[u];
        VOX_TRY(conv_gemm(st, uw.tconv, A, nullptr, slots, n, L, 0, off0, B, nullptr, nullptr, 0));      // B [n*2L, latent]
        L *= 2;
        hipLaunchKernelGGL(k_dwconv_ln, dim3(n * L), dim3(256), (size_t)LD * 4, st, B, m->st_dw[u], slots, uw.dw_w, uw.dw_b,
                           uw.ln_w, uw.ln_b, C, L, LD, 1e-6f);
        state_update(st, m->st_dw[u], slots, B, n, L, 6, LD);
        VOX_TRY(conv_gemm(st, uw.pw1, C, nullptr, slots, n, L, 0, off0, D, nullptr, nullptr, 1));        // GELU
        VOX_TRY(conv_gemm(st, uw.pw2, D, nullptr, slots, n, L, 0, off0, A, B, uw.gamma, 0));             // A = B + gamma*pw2
    }
    // ---- decoder ----
    const int off7[7] = {6, 5, 4, 3, 2, 1, 0};
    VOX_TRY(conv_gemm(st, w.dec0, A, m->st_dec0, slots, n, L, 6, off7, B, nullptr, nullptr, 0));          // B [nL, decoder_dim]
    state_update(st, m->st_dec0, slots, A, n, L, 6, LD);
    // SnakeBeta is fused into the epilogue of the conv that produces its input: a conv whose output is only ever consumed
    // through a snake (conv1 of a residual unit) writes just the activated tensor, the others write both (the plain tensor
    // is the residual branch).  t1 always holds the activated input of the next conv.
    // Two-term mode (planes == 2, the default): the activated tensors t1 / t3 — only ever read as GEMM operands — are kept as S2
    // rows (ConvGemmArgs), split once where they are produced; the residual stream h stays fp32.  Their history rows (st_tc /
    // st_ru) are byte copies of those rows, i.e. in the same format: change the mode only on freshly reset slots.
    const int s2 = m->planes == 2;
    float* h = B;      // current activations
    float* t1 = A;
    float* t2 = C;
    float* t3 = D;
    snake(st, h, w.blocks[0].snake0, t1, (size_t)n * L, c.decoder_dim);      // (dec0's input A is t1: no room for a fused second output)
    for (int b = 0; b < 4; ++b) {
        const vox_codec_block_w& bw = w.blocks[b];
        const int cin = c.decoder_dim >> b, cout = c.decoder_dim >> (b + 1), r = c.rates[b];
        const int offt[2] = {0, 1};
        // t2 [n*L*r, cout] = tconv(t1); t3 = act1 of unit 0 applied to it (t1 is still being read by other blocks)
        VOX_TRY(conv_gemm(st, bw.tconv, t1, m->st_tc[b], slots, n, L, 1, offt, t2, nullptr, nullptr, 0, t3, &bw.res[0].act1, cout,
                          nullptr, nullptr, nullptr, 0.0f, s2 && b > 0, s2));
        state_update(st, m->st_tc[b], slots, t1, n, L, 1, cin);
        L *= r;
        { float* x = h; h = t2; t2 = x; }
        { float* x = t1; t1 = t3; t3 = x; }             // t1 = act1(h)
        for (int u = 0; u < 3; ++u) {
            const vox_codec_res_w& rw = bw.res[u];
            const int d = u == 0 ? 1 : (u == 1 ? 3 : 9);
            const int offd[7] = {6 * d, 5 * d, 4 * d, 3 * d, 2 * d, d, 0};
            const vox_snake_w* nx = u < 2 ? &bw.res[u + 1].act1 : (b < 3 ? &w.blocks[b + 1].snake0 : &w.final_snake);
            if (s2) {
                // the whole unit in one launch where the tile covers all channels (k_res_unit): the activated intermediate stays on chip;
                // the next activation goes to t3 (neighbouring tiles still read their halo rows of t1), then the two swap
                int fused = 0;
                VOX_TRY(conv_unit(st, rw.conv1, rw.conv2, t1, m->st_ru[b][u], slots, n, L, 6 * d, offd, rw.act2, h, t3, nx, !(b == 3 && u == 2), &fused));
                if (fused) {
                    state_update(st, m->st_ru[b][u], slots, t1, n, L, 6 * d, cout);
                    { float* x = t1; t1 = t3; t3 = x; }
                    continue;
                }
            }
            // t3 = act2(conv1(t1))   (the plain conv1 output has no other consumer)
            VOX_TRY(conv_gemm(st, rw.conv1, t1, m->st_ru[b][u], slots, n, L, 6 * d, offd, nullptr, nullptr, nullptr, 0, t3, &rw.act2, cout,
                              nullptr, nullptr, nullptr, 0.0f, s2, s2));
            state_update(st, m->st_ru[b][u], slots, t1, n, L, 6 * d, cout);
            // h += conv2(t3); t1 = the next consumer's activation of the new h
            VOX_TRY(conv_gemm(st, rw.conv2, t3, nullptr, slots, n, L, 0, off0, h, h, nullptr, 0, t1, nx, cout,
                              nullptr, nullptr, nullptr, 0.0f, s2, s2 && !(b == 3 && u == 2)));      // (the final conv reads fp32 rows)
        }
    }
    const int cl = c.decoder_dim >> 4;
    if (L >= 512 && cl % 4 == 0 && (size_t)(134 * (cl + 1) + 7 * cl) * 4 <= 60 * 1024)
        hipLaunchKernelGGL(k_final_conv_rows, dim3((L + 127) / 128, n), dim3(128), (size_t)(134 * (cl + 1) + 7 * cl) * 4, st, t1, m->st_final,
                           slots, w.final_w, w.final_b, out, L, cl);
    else
        hipLaunchKernelGGL(k_final_conv, dim3((L + 3) / 4, n), dim3(256), 0, st, t1, m->st_final, slots, w.final_w, w.final_b, out,
                           L, cl);
    state_update(st, m->st_final, slots, t1, n, L, 6, cl);
    return VOX_OK;
}

}  // extern "C"

// ================================================================================================
// Mimi decoder (CSM's codec), stateless per chunk.  Same building blocks: implicit-GEMM convolutions on split-fp32
// MFMA (look-back rows before the chunk start read as zero = the reference's constant padding of a fresh state),
// transposed conv with stride r = 2-tap GEMM writing r*Cout contiguous values per input row (the reference trims the
// K-S rightmost samples: exactly the part the second tap of the NEXT chunk would add).
// ================================================================================================
// y[b, 2t+j, c] = x[b,t,c] * w[c][j] + x[b,t-1,c] * w[c][j+2]        (ConvTrUpsample1d, mimi.py:2272-2323)
// state (streaming): [slots][1][C] = the previous chunk's last input row; NULL = fresh state (zeros)
__global__ __launch_bounds__(256) void k_mimi_upsample(const float* x, const float* w, float* y, int L, int C, size_t total,
                                                        const float* state, const int* slots) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t row = i / C;                     // output row = b*2L + 2t + j
        const int tj = (int)(row % (2 * (size_t)L)), t = tj >> 1, j = tj & 1;
        const size_t b = row / (2 * (size_t)L);
        const float cur = x[(b * L + t) * C + c];
        const float prev = t > 0 ? x[(b * L + t - 1) * C + c] : (state ? state[(size_t)slots[b] * C + c] : 0.0f);
        y[i] = cur * w[c * 4 + j] + prev * w[c * 4 + j + 2];
    }
}
__global__ __launch_bounds__(256) void k_layernorm_f32(const float* x, const float* w, const float* b, float* y, int C, float eps) {
    __shared__ float red[8];
    const float* xr = x + (size_t)blockIdx.x * C;
    float s = 0.0f;
    for (int i = threadIdx.x; i < C; i += 256) s += xr[i];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    float v = 0.0f;
    for (int i = threadIdx.x; i < C; i += 256) { const float d = xr[i] - mean; v += d * d; }
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = v;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)C + eps);
    for (int i = threadIdx.x; i < C; i += 256) y[(size_t)blockIdx.x * C + i] = (xr[i] - mean) * rstd * w[i] + b[i];
}
__global__ __launch_bounds__(256) void k_elu(const float* x, float* y, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        y[i] = v > 0.0f ? v : expm1f(v);
    }
}
// causal attention inside one chunk of L <= 64 tokens with RoPE on interleaved pairs at positions 0..L-1
// (apply_rope mimi.py:874-931; offset 0: stateless).  qkv rows [3][H][D]; one block per (head, request).
__global__ __launch_bounds__(256) void k_mimi_attn(const float* qkv, float* out, int L, int H, int D, int context, float max_period) {
    extern __shared__ float sm[];   // Q [L][D], K [L][D+1], V [L][D+1], P [4][L]
    const int LD = D + 1;
    float* Qs = sm;
    float* Ks = Qs + (size_t)L * D;
    float* Vs = Ks + (size_t)L * LD;
    float* Ps = Vs + (size_t)L * LD;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HD = H * D;
    for (int e = tid; e < L * (D / 2); e += 256) {
        const int t = e / (D / 2), i = e % (D / 2);
        const float* r = qkv + ((size_t)(b * L + t)) * 3 * HD + (size_t)h * D + 2 * i;
        const float freq = expf((float)i * (-logf(max_period) * 2.0f / (float)D));
        const float ang = freq * (float)t;
        const float cr = cosf(ang), ci = sinf(ang);
        const float qr = r[0], qi = r[1], kr = r[HD], ki = r[HD + 1];
        Qs[t * D + 2 * i] = qr * cr - qi * ci;
        Qs[t * D + 2 * i + 1] = qr * ci + qi * cr;
        Ks[t * LD + 2 * i] = kr * cr - ki * ci;
        Ks[t * LD + 2 * i + 1] = kr * ci + ki * cr;
        Vs[t * LD + 2 * i] = r[2 * HD];
        Vs[t * LD + 2 * i + 1] = r[2 * HD + 1];
    }
    __syncthreads();
    const float scale = rsqrtf((float)D);
    float* P = Ps + (size_t)wave * L;
    for (int i = wave; i < L; i += 4) {
        const float* q = Qs + (size_t)i * D;
        const int j = lane;
        const bool vis = j < L && j <= i && (i - j) < context;
        float sc = -INFINITY;
        if (vis) {
            sc = 0.0f;
            for (int d = 0; d < D; ++d) sc = fmaf(q[d], Ks[j * LD + d], sc);
            sc *= scale;
        }
        float mx = sc;
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const float pj = vis ? expf(sc - mx) : 0.0f;
        float ls = pj;
        for (int off = 32; off >= 1; off >>= 1) ls += __shfl_xor(ls, off, 64);
        if (j < L) P[j] = pj;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (int d = lane; d < D; d += 64) {
            float o = 0.0f;
            for (int jj = 0; jj <= i; ++jj) o = fmaf(P[jj], Vs[jj * LD + d], o);
            out[((size_t)(b * L + i)) * HD + (size_t)h * D + d] = o / ls;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}
// Streaming variant: queries = the L new rows at absolute positions pos0 .. pos0+L-1 of the slot, keys = the slot's K/V ring
// (post-RoPE K) + the new rows; visibility 0 <= p_q - p_k < context.  Ring capacity RC >= context + L - 1 so that a new key
// never overwrites one an earlier query of the same chunk still sees.  One block per (head, request): the visible K window
// (<= context + L - 1 rows) sits in LDS (row stride D+1), V is read from the ring (coalesced rows, L2-resident).
__global__ __launch_bounds__(256) void k_mimi_attn_stream(const float* qkv, float* out, float* kring, float* vring, const long* pos,
                                                           const int* slots, int L, int H, int D, int context, int RC, float max_period) {
    extern __shared__ float sm[];   // Q [L][D], K [context+L][D+1], P [4][context+L]
    const int LD = D + 1, W = context + L;
    float* Qs = sm;
    float* Ks = Qs + (size_t)L * D;
    float* Ps = Ks + (size_t)W * LD;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HD = H * D;
    const long p0 = pos[slots[b]];
    float* kr = kring + (size_t)slots[b] * RC * HD + (size_t)h * D;     // row stride HD
    float* vr = vring + (size_t)slots[b] * RC * HD + (size_t)h * D;
    for (int e = tid; e < L * (D / 2); e += 256) {       // RoPE the new q / k at their absolute positions, append k / v
        const int t = e / (D / 2), i = e % (D / 2);
        const float* r = qkv + ((size_t)(b * L + t)) * 3 * HD + (size_t)h * D + 2 * i;
        const float freq = expf((float)i * (-logf(max_period) * 2.0f / (float)D));
        const float ang = freq * (float)(p0 + t);
        const float cr = cosf(ang), ci = sinf(ang);
        const float qr = r[0], qi = r[1], k0 = r[HD], k1 = r[HD + 1];
        Qs[t * D + 2 * i] = qr * cr - qi * ci;
        Qs[t * D + 2 * i + 1] = qr * ci + qi * cr;
        const size_t ro = (size_t)((p0 + t) % RC) * HD + 2 * i;
        kr[ro] = k0 * cr - k1 * ci;
        kr[ro + 1] = k0 * ci + k1 * cr;
        vr[ro] = r[2 * HD];
        vr[ro + 1] = r[2 * HD + 1];
    }
    __syncthreads();                                       // (block-scope: this block is the only reader of its ring rows)
    const long lo = (p0 - context + 1) > 0 ? (p0 - context + 1) : 0;     // oldest position any query of the chunk sees
    const int nk = (int)(p0 + L - lo);                     // window rows [lo, p0+L)
    for (int e = tid; e < nk * D; e += 256) {
        const int j = e / D, d = e % D;
        Ks[j * LD + d] = kr[(size_t)((lo + j) % RC) * HD + d];
    }
    __syncthreads();
    const float scale = rsqrtf((float)D);
    float* P = Ps + (size_t)wave * W;
    for (int i = wave; i < L; i += 4) {
        const float* q = Qs + (size_t)i * D;
        const long pq = p0 + i;
        float mx = -INFINITY;
        for (int j = lane; j < nk; j += 64) {
            const long pk = lo + j;
            float sc = -INFINITY;
            if (pk <= pq && (pq - pk) < context) {
                sc = 0.0f;
                for (int d = 0; d < D; ++d) sc = fmaf(q[d], Ks[j * LD + d], sc);
                sc *= scale;
            }
            P[j] = sc;
            mx = fmaxf(mx, sc);
        }
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float ls = 0.0f;
        for (int j = lane; j < nk; j += 64) {
            const float pj = P[j] == -INFINITY ? 0.0f : expf(P[j] - mx);
            P[j] = pj;
            ls += pj;
        }
        for (int off = 32; off >= 1; off >>= 1) ls += __shfl_xor(ls, off, 64);
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        const int jlo = (int)((pq - context + 1 > lo ? pq - context + 1 : lo) - lo), jhi = (int)(pq - lo);
        for (int d = lane; d < D; d += 64) {
            float o = 0.0f;
            for (int jj = jlo; jj <= jhi; ++jj) o = fmaf(P[jj], vr[(size_t)((lo + jj) % RC) * HD + d], o);
            out[((size_t)(b * L + i)) * HD + (size_t)h * D + d] = o / ls;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}
// last conv: C -> 1 channel, kernel K, zero history
__global__ __launch_bounds__(256) void k_mimi_final(const float* x, const float* w, float bias, float* out, int L, int C, int K,
                                                     const float* state, const int* slots) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= L) return;
    float acc = 0.0f;
    for (int e = lane; e < K * C; e += 64) {
        const int k = e / C, c = e % C;
        const int r = t - (K - 1) + k;
        if (r >= 0) acc += w[c * K + k] * x[((size_t)b * L + r) * C + c];
        else if (state) acc += w[c * K + k] * state[((size_t)slots[b] * (K - 1) + (K - 1 + r)) * C + c];   // streaming history
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[(size_t)b * L + t] = acc + bias;
}

struct vox_mimi {
    vox_ctx* ctx;
    vox_mimi_config cfg;
    vox_mimi_weights w;
    int max_batch, max_frames;
    float *q0, *qr, *buf[4];
    int32_t* zero_slots;
    // streaming state (vox_mimi_stream_enable), per slot: the look-back rows of every causal conv / transposed conv input,
    // a K/V ring per transformer layer and the number of transformer-rate rows seen so far
    int max_slots = 0, ring_cap = 0;
    float *st_up = nullptr, *st_dec0 = nullptr, *st_tc[4] = {nullptr, nullptr, nullptr, nullptr},
          *st_c1[4] = {nullptr, nullptr, nullptr, nullptr}, *st_final = nullptr, *kring = nullptr, *vring = nullptr;
    long* pos = nullptr;
};

static void elu(hipStream_t st, const float* x, float* y, size_t total) {
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_elu, dim3(grid), dim3(256), 0, st, x, y, total);
}

extern "C" {

int vox_mimi_create(vox_ctx* ctx, const vox_mimi_config* cfg, const vox_mimi_weights* w, int max_batch, int max_frames,
                    vox_mimi** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "mimi_create: NULL");
    if (cfg->num_layers > 16 || cfg->dim % cfg->num_heads || 2 * max_frames > 64)
        return vox_fail(VOX_ERR_INVALID, "mimi_create: <= 16 layers, <= 32 frames per chunk");
    vox_mimi* m = new vox_mimi();
    m->ctx = ctx; m->cfg = *cfg; m->w = *w; m->max_batch = max_batch; m->max_frames = max_frames;
    // largest activation: rows x channels is the same (2T * 16 n_filters * prod so far / 2^i) at every SEANet stage up to
    // the ratio products; bound it by the widest of each stage
    size_t L = (size_t)2 * max_frames, ch = (size_t)16 * cfg->n_filters, maxf = L * ch;
    for (int i = 0; i < 4; ++i) {
        L *= cfg->ratios[i];
        ch /= 2;
        maxf = maxf > L * ch ? maxf : L * ch;
    }
    const size_t tr = (size_t)2 * max_frames * (3 * cfg->dim > cfg->ffn ? 3 * cfg->dim : cfg->ffn);
    maxf = (maxf > tr ? maxf : tr) * max_batch;
    bool ok = hipMalloc((void**)&m->q0, (size_t)max_batch * max_frames * cfg->vq_dim * 4) == hipSuccess &&
              hipMalloc((void**)&m->qr, (size_t)max_batch * max_frames * cfg->vq_dim * 4) == hipSuccess &&
              hipMalloc((void**)&m->zero_slots, (size_t)max_batch * 4) == hipSuccess;
    for (int i = 0; i < 4; ++i) ok = ok && hipMalloc((void**)&m->buf[i], maxf * 4) == hipSuccess;
    if (!ok) return vox_fail(VOX_ERR_NOMEM, "mimi_create: hipMalloc failed");
    VOX_HIP(hipMemset(m->zero_slots, 0, (size_t)max_batch * 4));
    *out = m;
    return VOX_OK;
}
void vox_mimi_destroy(vox_mimi* m) {
    if (!m) return;
    (void)hipFree(m->q0); (void)hipFree(m->qr); (void)hipFree(m->zero_slots);
    (void)hipFree(m->st_up); (void)hipFree(m->st_dec0); (void)hipFree(m->st_final); (void)hipFree(m->kring); (void)hipFree(m->vring); (void)hipFree(m->pos);
    for (int b = 0; b < 4; ++b) { (void)hipFree(m->st_tc[b]); (void)hipFree(m->st_c1[b]); }
    for (int i = 0; i < 4; ++i) (void)hipFree(m->buf[i]);
    delete m;
}
}  // extern "C"

// slots == NULL: stateless chunk (fresh state: the reference's MimiDecoder.decode); else streaming on the given state slots
static int mimi_run(vox_mimi* m, hipStream_t st, const int32_t* codes, int code_stride, const int32_t* slots, int n, int T, float* out) {
    const vox_mimi_config& c = m->cfg;
    const vox_mimi_weights& w = m->w;
    const int C = c.dim, H = c.num_heads, D = c.dim / c.num_heads;
    const bool S = slots != nullptr;
    const int* sl = S ? slots : m->zero_slots;
    float *A = m->buf[0], *B = m->buf[1], *Cb = m->buf[2], *Db = m->buf[3];
    const int off0[1] = {0};
    int L = T;
    hipLaunchKernelGGL(k_rvq, dim3(n * L), dim3(256), 0, st, codes, code_stride, w.emb, c.n_q, c.bins, c.vq_dim, m->q0, m->qr);
    VOX_TRY(conv_gemm(st, w.rvq_first_out, m->q0, nullptr, sl, n, L, 0, off0, A, nullptr, nullptr, 0));
    VOX_TRY(conv_gemm(st, w.rvq_rest_out, m->qr, nullptr, sl, n, L, 0, off0, B, A, nullptr, 0));            // B [nL, dim]
    {
        const size_t total = (size_t)n * 2 * L * C;
        int grid = (int)((total + 255) / 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(k_mimi_upsample, dim3(grid), dim3(256), 0, st, B, w.up_w, A, L, C, total, S ? m->st_up : nullptr, sl);   // A [n*2L, dim]
        if (S) state_update(st, m->st_up, sl, B, n, L, 1, C);
    }
    L *= 2;
    for (int l = 0; l < c.num_layers; ++l) {                                                                 // h = A
        const vox_mimi_layer_w& lw = w.layers[l];
        hipLaunchKernelGGL(k_layernorm_f32, dim3(n * L), dim3(256), 0, st, A, lw.ln1_w, lw.ln1_b, Cb, C, c.ln_eps);
        VOX_TRY(conv_gemm(st, lw.qkv, Cb, nullptr, sl, n, L, 0, off0, Db, nullptr, nullptr, 0));             // Db [nL, 3*dim]
        if (S) {
            const size_t ring = (size_t)m->max_slots * m->ring_cap * C;
            const size_t smem = ((size_t)L * D + (size_t)(c.context + L) * (D + 1) + 4 * (size_t)(c.context + L)) * 4;
            hipLaunchKernelGGL(k_mimi_attn_stream, dim3(H, n), dim3(256), smem, st, Db, Cb, m->kring + l * ring, m->vring + l * ring,
                               m->pos, sl, L, H, D, c.context, m->ring_cap, c.max_period);
        } else {
            hipLaunchKernelGGL(k_mimi_attn, dim3(H, n), dim3(256), (size_t)(L * D + 2 * L * (D + 1) + 4 * L) * 4, st, Db, Cb, L, H, D,
                               c.context, c.max_period);
        }
        VOX_TRY(conv_gemm(st, lw.o, Cb, nullptr, sl, n, L, 0, off0, A, A, lw.scale1, 0));                    // h += s1 * o(attn)
        hipLaunchKernelGGL(k_layernorm_f32, dim3(n * L), dim3(256), 0, st, A, lw.ln2_w, lw.ln2_b, Cb, C, c.ln_eps);
        VOX_TRY(conv_gemm(st, lw.fc1, Cb, nullptr, sl, n, L, 0, off0, Db, nullptr, nullptr, 1));             // GELU
        VOX_TRY(conv_gemm(st, lw.fc2, Db, nullptr, sl, n, L, 0, off0, A, A, lw.scale2, 0));                  // h += s2 * mlp
    }
    if (S) hipLaunchKernelGGL(k_pos_advance, dim3((n + 255) / 256), dim3(256), 0, st, m->pos, sl, n, L);
    // ---- SEANet decoder ----
    int offk[CG_MAXTAPS];
    for (int k = 0; k < c.kernel_size; ++k) offk[k] = c.kernel_size - 1 - k;
    const int Pk = c.kernel_size - 1;
    VOX_TRY(conv_gemm(st, w.dec0, A, S ? m->st_dec0 : nullptr, sl, n, L, S ? Pk : 0, offk, B, nullptr, nullptr, 0));   // B [nL, 16 nf]
    if (S) state_update(st, m->st_dec0, sl, A, n, L, Pk, C);
    float *h = B, *t1 = A, *t2 = Cb, *t3 = Db;
    int ch = 16 * c.n_filters;
    for (int b = 0; b < 4; ++b) {
        const vox_mimi_block_w& bw = w.blocks[b];
        const int r = c.ratios[b];
        elu(st, h, t1, (size_t)n * L * ch);
        const int offt[2] = {0, 1};
        VOX_TRY(conv_gemm(st, bw.tconv, t1, S ? m->st_tc[b] : nullptr, sl, n, L, S ? 1 : 0, offt, t2, nullptr, nullptr, 0));   // t2 [n*L*r, ch/2]
        if (S) state_update(st, m->st_tc[b], sl, t1, n, L, 1, ch);
        L *= r;
        ch /= 2;
        { float* x = h; h = t2; t2 = x; }
        elu(st, h, t1, (size_t)n * L * ch);
        const int off3[3] = {2, 1, 0};
        VOX_TRY(conv_gemm(st, bw.conv1, t1, S ? m->st_c1[b] : nullptr, sl, n, L, S ? 2 : 0, off3, t3, nullptr, nullptr, 0));   // t3 [nL, ch/2]
        if (S) state_update(st, m->st_c1[b], sl, t1, n, L, 2, ch);
        elu(st, t3, t1, (size_t)n * L * (ch / 2));
        VOX_TRY(conv_gemm(st, bw.conv2, t1, nullptr, sl, n, L, 0, off0, h, h, nullptr, 0));                  // h += conv2(...)
    }
    elu(st, h, t1, (size_t)n * L * ch);
    hipLaunchKernelGGL(k_mimi_final, dim3((L + 3) / 4, n), dim3(256), 0, st, t1, w.final_w, w.final_b, out, L, ch, c.last_kernel_size,
                       S ? m->st_final : nullptr, sl);
    if (S) state_update(st, m->st_final, sl, t1, n, L, c.last_kernel_size - 1, ch);
    return VOX_OK;
}

extern "C" {

int vox_mimi_decode(vox_mimi* m, void* stream, const int32_t* codes, int code_stride, int n, int T, float* out) {
    if (!m || !codes || !out) return vox_fail(VOX_ERR_INVALID, "mimi_decode: NULL");
    if (n < 1 || n > m->max_batch || T < 1 || T > m->max_frames) return vox_fail(VOX_ERR_INVALID, "mimi_decode: n=%d T=%d out of range", n, T);
    return mimi_run(m, (hipStream_t)stream, codes, code_stride, nullptr, n, T, out);
}

int vox_mimi_stream_enable(vox_mimi* m, int max_slots) {
    if (!m || max_slots < 1 || m->max_slots) return vox_fail(VOX_ERR_INVALID, "mimi_stream_enable: bad argument or already enabled");
    const vox_mimi_config& c = m->cfg;
    if (c.kernel_size - 1 > 8 || c.last_kernel_size < 2) return vox_fail(VOX_ERR_INVALID, "mimi_stream_enable: unsupported kernel sizes");
    const size_t S = max_slots, C = c.dim;
    m->ring_cap = c.context + 2 * m->max_frames;           // >= context + L - 1
    bool ok = hipMalloc((void**)&m->st_up, S * C * 4) == hipSuccess && hipMalloc((void**)&m->st_dec0, S * (c.kernel_size - 1) * C * 4) == hipSuccess &&
              hipMalloc((void**)&m->kring, (size_t)c.num_layers * S * m->ring_cap * C * 4) == hipSuccess &&
              hipMalloc((void**)&m->vring, (size_t)c.num_layers * S * m->ring_cap * C * 4) == hipSuccess &&
              hipMalloc((void**)&m->pos, S * sizeof(long)) == hipSuccess;
    size_t ch = (size_t)16 * c.n_filters;
    for (int b = 0; ok && b < 4; ++b) {
        ok = hipMalloc((void**)&m->st_tc[b], S * ch * 4) == hipSuccess && hipMalloc((void**)&m->st_c1[b], S * 2 * (ch / 2) * 4) == hipSuccess;
        ch /= 2;
    }
    ok = ok && hipMalloc((void**)&m->st_final, S * (c.last_kernel_size - 1) * ch * 4) == hipSuccess;
    if (!ok) return vox_fail(VOX_ERR_NOMEM, "mimi_stream_enable: hipMalloc failed");
    m->max_slots = max_slots;
    for (int sl = 0; sl < max_slots; ++sl) VOX_TRY(vox_mimi_reset_slot(m, nullptr, sl));
    VOX_HIP(hipDeviceSynchronize());
    return VOX_OK;
}

int vox_mimi_reset_slot(vox_mimi* m, void* stream, int slot) {
    if (!m || slot < 0 || slot >= m->max_slots) return vox_fail(VOX_ERR_INVALID, "mimi_reset_slot: slot %d out of range", slot);
    hipStream_t st = (hipStream_t)stream;
    const vox_mimi_config& c = m->cfg;
    const size_t C = c.dim;
    VOX_HIP(hipMemsetAsync(m->st_up + (size_t)slot * C, 0, C * 4, st));
    VOX_HIP(hipMemsetAsync(m->st_dec0 + (size_t)slot * (c.kernel_size - 1) * C, 0, (c.kernel_size - 1) * C * 4, st));
    size_t ch = (size_t)16 * c.n_filters;
    for (int b = 0; b < 4; ++b) {
        VOX_HIP(hipMemsetAsync(m->st_tc[b] + (size_t)slot * ch, 0, ch * 4, st));
        VOX_HIP(hipMemsetAsync(m->st_c1[b] + (size_t)slot * 2 * (ch / 2), 0, 2 * (ch / 2) * 4, st));
        ch /= 2;
    }
    VOX_HIP(hipMemsetAsync(m->st_final + (size_t)slot * (c.last_kernel_size - 1) * ch, 0, (c.last_kernel_size - 1) * ch * 4, st));
    VOX_HIP(hipMemsetAsync(m->pos + slot, 0, sizeof(long), st));     // (the K/V ring needs no clearing: only positions < pos are read)
    return VOX_OK;
}

int vox_mimi_decode_chunk(vox_mimi* m, void* stream, const int32_t* codes, int code_stride, const int32_t* slots, int n, int T, float* out) {
    if (!m || !codes || !out || !slots) return vox_fail(VOX_ERR_INVALID, "mimi_decode_chunk: NULL");
    if (!m->max_slots) return vox_fail(VOX_ERR_INVALID, "mimi_decode_chunk: call vox_mimi_stream_enable first");
    if (n < 1 || n > m->max_batch || T < 1 || T > m->max_frames) return vox_fail(VOX_ERR_INVALID, "mimi_decode_chunk: n=%d T=%d out of range", n, T);
    return mimi_run(m, (hipStream_t)stream, codes, code_stride, slots, n, T, out);
}

}  // extern "C"


// ====================================================================================================================
// SNAC decoder (Orpheus): SNAC.decode of /root/reference/vox_serve/tokenizer/snac.py:438-441, stateless per window.
// Layout as above (fp32, time-major rows).  A transposed conv with stride r, kernel 2r, padding r/2 is the 2-tap GEMM
//   G[q][j*Cout + co] = x[q] . W[:, co, j] + x[q-1] . W[:, co, j + r],   q = 0..Tin  (x[-1] = x[Tin] = 0),
// and the conv's output row o is row o + r/2 of G seen as [(Tin + 1) * r][Cout].  Instead of copying, every later kernel
// of the block works on that padded block (Lb = (Tin + 1) r rows per request, valid rows [v0, v0 + Lv)): the 1x1 GEMMs run
// on all rows (rows never mix), the depthwise convs read zeros outside the valid range, and the next block's Snake
// compacts the valid rows again (plus the zero row x[Tin]).
// ====================================================================================================================
struct SnacGeo { int B, Lb, v0, Lv, C; };

__device__ __forceinline__ float snake_a(float v, float alpha, float invb) { return snake_f(v, alpha, invb); }

// z[b][t][c] = sum_i tab_i[code_i[b][t / stride_i]][c]  (sequential in i; from_codes: snac.py:350-357)
__global__ __launch_bounds__(256) void k_snac_embed(const int* codes, int code_row, const float* t0, const float* t1, const float* t2,
                                                     const float* t3, int n_levels, int s0, int s1, int s2, int s3, int bins, int T,
                                                     int C, float* z) {
    const int b = blockIdx.z, t = blockIdx.y;
    const float* tabs[4] = {t0, t1, t2, t3};
    const int st[4] = {s0, s1, s2, s3};
    const int* cr = codes + (size_t)b * code_row;
    int id[4], off = 0;
    for (int i = 0; i < n_levels; ++i) {
        int v = cr[off + t / st[i]];
        id[i] = v < 0 ? 0 : (v >= bins ? bins - 1 : v);
        off += T / st[i];
    }
    for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
        float acc = tabs[0][(size_t)id[0] * C + c];
        for (int i = 1; i < n_levels; ++i) acc = acc + tabs[i][(size_t)id[i] * C + c];
        z[((size_t)b * T + t) * C + c] = acc;
    }
}

// depthwise conv k = 7, dilation d, "same" zero padding inside the valid rows, optional Snake on the input taps and on
// the output:  y[t][c] = post( bias[c] + sum_j w[c][j] * pre(x[t + (j - 3) d][c]) )      (ResidualUnit: snac.py:160-176)
__global__ __launch_bounds__(256) void k_snac_dw(const float* x, const float* w, const float* bias, const float* pa, const float* pib,
                                                  const float* qa, const float* qib, float* y, SnacGeo g, int dil) {
    const size_t total = (size_t)g.B * g.Lb * g.C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        const size_t row = i / g.C;
        const int t = (int)(row % g.Lb) - g.v0;
        float v = 0.0f;
        if (t >= 0 && t < g.Lv) {
            const float a = pa ? pa[c] : 0.0f, ib = pa ? pib[c] : 0.0f;
            v = bias[c];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int tt = t + (j - 3) * dil;
                if (tt >= 0 && tt < g.Lv) {
                    float xv = x[(row + (size_t)((j - 3) * dil)) * g.C + c];
                    if (pa) xv = snake_a(xv, a, ib);
                    v = fmaf(w[c * 7 + j], xv, v);
                }
            }
            if (qa) v = snake_a(v, qa[c], qib[c]);
        }
        y[i] = v;
    }
}

// Snake of the valid rows -> compact [B][Lv + 1][C] with a trailing zero row (the x[Tin] = 0 of the transposed conv)
__global__ __launch_bounds__(256) void k_snac_snake_pack(const float* x, const float* alpha, const float* invb, float* y, SnacGeo g) {
    const size_t total = (size_t)g.B * (g.Lv + 1) * g.C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        const size_t row = i / g.C;
        const int b = (int)(row / (g.Lv + 1)), t = (int)(row % (g.Lv + 1));
        y[i] = t < g.Lv ? snake_a(x[((size_t)b * g.Lb + g.v0 + t) * g.C + c], alpha[c], invb[c]) : 0.0f;
    }
}

__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* o0,
                                        uint32_t* o1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    *o0 = c0; *o1 = c1;
}
// per-row NoiseBlock multiplier in the padded geometry (junk rows 0): given noise [B][Lv] or the seeded Philox stream
__global__ __launch_bounds__(256) void k_snac_noise(const float* noise, uint64_t seed, const uint32_t* stream_base, int stage, int n_stages,
                                                     float* rs, SnacGeo g) {
    const size_t total = (size_t)g.B * g.Lb;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / g.Lb), t = (int)(i % g.Lb) - g.v0;
        float v = 0.0f;
        if (t >= 0 && t < g.Lv) {
            if (noise) v = noise[(size_t)b * g.Lv + t];
            else {
                const uint32_t stream = (stream_base ? stream_base[b] : (uint32_t)(b * n_stages)) + (uint32_t)stage;
                uint32_t w0, w1;
                philox4((uint32_t)t, stream, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), &w0, &w1);
                const float u1 = ((float)(w0 >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);
                v = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
            }
        }
        rs[i] = v;
    }
}

// final Snake -> conv k7 (C -> 1) -> tanh, samples [out_off, out_off + out_len) of the valid rows; one wave per sample
__global__ __launch_bounds__(256) void k_snac_final(const float* x, const float* alpha, const float* invb, const float* w, float bias,
                                                     float* out, SnacGeo g, int out_off, int out_len) {
    const int lane = threadIdx.x & 63;
    const size_t s = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= (size_t)g.B * out_len) return;
    const int b = (int)(s / out_len), t = out_off + (int)(s % out_len);
    float acc = 0.0f;
    for (int c = lane; c < g.C; c += 64) {
        const float a = alpha[c], ib = invb[c];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int tt = t + j - 3;
            if (tt >= 0 && tt < g.Lv) acc = fmaf(w[c * 7 + j], snake_a(x[((size_t)b * g.Lb + g.v0 + tt) * g.C + c], a, ib), acc);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[s] = tanhf(acc + bias);
}

// Snake of the valid rows, zeros elsewhere (same padded geometry): the input of a DENSE k7 conv, whose taps must read zeros
// outside the sequence                                                                        (ResidualUnit with groups = 1)
__global__ __launch_bounds__(256) void k_snac_snake_mask(const float* x, const float* alpha, const float* invb, float* y, SnacGeo g) {
    const size_t total = (size_t)g.B * g.Lb * g.C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        const int t = (int)((i / g.C) % g.Lb) - g.v0;
        y[i] = (t >= 0 && t < g.Lv) ? snake_a(x[i], alpha[c], invb[c]) : 0.0f;
    }
}

// LocalMHA core (snac.py:33-47, 76-90): one block per (window, head, request).  qkv rows [T][3C] (q | k | v, heads of 64 inside
// each), window = ws consecutive frames; q and k get the rotary embedding of their position INSIDE the window (no xpos: scale 1),
// full (non-causal) softmax over the window, scale 1/8.  out rows [T][C].
__global__ __launch_bounds__(256) void k_snac_local_attn(const float* qkv, float* out, int T, int C, int ws) {
    constexpr int DH = 64;
    __shared__ float q[32 * DH], k[32 * DH], v[32 * DH], sc[32 * 65];      // ws <= 32 (vox_snac_create)
    const int w = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const float* base = qkv + ((size_t)b * T + (size_t)w * ws) * 3 * C + h * DH;
    for (int e = tid; e < ws * DH; e += 256) {
        const int n = e / DH, d = e % DH, dp = d < DH / 2 ? d + DH / 2 : d - DH / 2;
        const float* row = base + (size_t)n * 3 * C;
        const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * (d % (DH / 2))) / (float)DH);
        const float fr = (float)n * inv_freq, cs = cosf(fr), sn = sinf(fr);
        const float sgn = d < DH / 2 ? -1.0f : 1.0f;                      // rotate_half: (-x2, x1)
        q[e] = row[d] * cs + sgn * row[dp] * sn;
        k[e] = row[C + d] * cs + sgn * row[C + dp] * sn;
        v[e] = row[2 * C + d];
    }
    __syncthreads();
    for (int e = tid; e < ws * ws; e += 256) {
        const int i = e / ws, j = e % ws;
        float s = 0.0f;
        for (int d = 0; d < DH; ++d) s = fmaf(q[i * DH + d], k[j * DH + d], s);
        sc[i * 65 + j] = s * 0.125f;
    }
    __syncthreads();
    if (tid < ws) {
        float mx = -INFINITY, l = 0.0f;
        for (int j = 0; j < ws; ++j) mx = fmaxf(mx, sc[tid * 65 + j]);
        for (int j = 0; j < ws; ++j) { const float p = expf(sc[tid * 65 + j] - mx); sc[tid * 65 + j] = p; l += p; }
        const float il = 1.0f / l;
        for (int j = 0; j < ws; ++j) sc[tid * 65 + j] *= il;
    }
    __syncthreads();
    for (int e = tid; e < ws * DH; e += 256) {
        const int i = e / DH, d = e % DH;
        float o = 0.0f;
        for (int j = 0; j < ws; ++j) o = fmaf(sc[i * 65 + j], v[j * DH + d], o);
        out[((size_t)b * T + (size_t)w * ws + i) * C + h * DH + d] = o;
    }
}

struct vox_snac {
    vox_ctx* ctx;
    vox_snac_config cfg;
    vox_snac_weights w;
    int max_batch, max_T, n_stages;
    float *buf[4], *rs;
    size_t buf_floats;
};

static inline int ew_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

extern "C" {

int vox_snac_create(vox_ctx* ctx, const vox_snac_config* cfg, const vox_snac_weights* w, int max_batch, int max_T, vox_snac** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "snac_create: NULL");
    if (cfg->n_levels < 1 || cfg->n_levels > 4 || max_batch < 1 || max_T < 1) return vox_fail(VOX_ERR_INVALID, "snac_create: bad config");
    vox_snac* m = new vox_snac();
    m->ctx = ctx; m->cfg = *cfg; m->w = *w; m->max_batch = max_batch; m->max_T = max_T;
    m->n_stages = 0;
    int ch = cfg->decoder_dim;
    if (cfg->attn_window < 0 || cfg->attn_window > 32 || (cfg->attn_window && cfg->decoder_dim % 64))
        return vox_fail(VOX_ERR_INVALID, "snac_create: attention window 1..32, decoder_dim a multiple of 64");
    size_t worst = (size_t)max_T * (cfg->latent_dim > cfg->decoder_dim ? cfg->latent_dim : cfg->decoder_dim) * (cfg->attn_window ? 3 : 1);
    size_t T = max_T, rs_rows = 0;
    for (int i = 0; i < 4 && cfg->rates[i] > 0; ++i) {
        const int r = cfg->rates[i];
        if (r % 2 || ch % 2) { delete m; return vox_fail(VOX_ERR_INVALID, "snac_create: rates must be even"); }
        const size_t Lb = (T + 1) * r;
        if (Lb * (ch / 2) > worst) worst = Lb * (ch / 2);
        if ((T + 1) * ch > worst) worst = (T + 1) * ch;
        if (Lb > rs_rows) rs_rows = Lb;
        T *= r; ch /= 2;
        ++m->n_stages;
    }
    m->buf_floats = worst * max_batch;
    bool ok = hipMalloc((void**)&m->rs, rs_rows * max_batch * 4) == hipSuccess;
    for (int i = 0; i < 4; ++i) ok = ok && hipMalloc((void**)&m->buf[i], m->buf_floats * 4) == hipSuccess;
    if (!ok) { vox_snac_destroy(m); return vox_fail(VOX_ERR_NOMEM, "snac_create: hipMalloc failed"); }
    *out = m;
    return VOX_OK;
}

void vox_snac_destroy(vox_snac* m) {
    if (!m) return;
    for (int i = 0; i < 4; ++i) (void)hipFree(m->buf[i]);
    (void)hipFree(m->rs);
    delete m;
}

int vox_snac_decode(vox_snac* m, void* stream, const int32_t* codes, int n, int T, const float* noise, uint64_t seed,
                    const uint32_t* stream_base, float* out, int out_off, int out_len) {
    if (!m || !codes || !out) return vox_fail(VOX_ERR_INVALID, "snac_decode: NULL");
    const vox_snac_config& c = m->cfg;
    if (n < 1 || n > m->max_batch || T < 1 || T > m->max_T || T % c.vq_strides[0])
        return vox_fail(VOX_ERR_INVALID, "snac_decode: n %d / T %d out of range", n, T);
    hipStream_t st = (hipStream_t)stream;
    int code_row = 0;
    for (int i = 0; i < c.n_levels; ++i) code_row += T / c.vq_strides[i];
    float *x = m->buf[0], *y = m->buf[1], *u = m->buf[2], *pk = m->buf[3];
    const vox_snac_weights& w = m->w;
    // latent: from_codes -> depthwise k7 -> 1x1
    hipLaunchKernelGGL(k_snac_embed, dim3((c.latent_dim + 255) / 256, T, n), dim3(256), 0, st, codes, code_row, w.tab[0], w.tab[1], w.tab[2],
                       w.tab[3], c.n_levels, c.vq_strides[0], c.vq_strides[1] ? c.vq_strides[1] : 1, c.vq_strides[2] ? c.vq_strides[2] : 1,
                       c.vq_strides[3] ? c.vq_strides[3] : 1, c.codebook_size, T, c.latent_dim, x);
    SnacGeo g{n, T, 0, T, c.latent_dim};
    static const int off1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, off2[8] = {0, 1, 0, 1, 0, 0, 0, 0};
    if (w.conv0.n_taps > 0) {      // non-depthwise: one dense k7 conv ("same" padding: tap j reads row t + j - 3), 2 weight planes
        static const int off7[14] = {3, 2, 1, 0, -1, -2, -3, 3, 2, 1, 0, -1, -2, -3};
        VOX_TRY(conv_gemm(st, w.conv0, x, nullptr, nullptr, n, T, 0, off7, y, nullptr, nullptr, 0));
        float* t = x; x = y; y = t;
    } else {
        hipLaunchKernelGGL(k_snac_dw, dim3(ew_grid((size_t)n * T * c.latent_dim)), dim3(256), 0, st, x, w.dw0_w, w.dw0_b, nullptr, nullptr,
                           nullptr, nullptr, y, g, 1);
        VOX_TRY(conv_gemm(st, w.pw0, y, nullptr, nullptr, n, T, 0, off1, x, nullptr, nullptr, 0));
    }
    g.C = c.decoder_dim;
    if (c.attn_window > 0) {       // LocalMHA: x += to_out(attention(rotary(to_qkv(LayerNorm(x)))))
        if (T % c.attn_window) return vox_fail(VOX_ERR_INVALID, "snac_decode: %d frames are not a multiple of the attention window %d", T, c.attn_window);
        hipLaunchKernelGGL(k_layernorm_f32, dim3(n * T), dim3(256), 0, st, x, w.attn_ln_w, w.attn_ln_b, y, g.C, 1e-5f);
        VOX_TRY(conv_gemm(st, w.attn_qkv, y, nullptr, nullptr, n, T, 0, off1, u, nullptr, nullptr, 0));          // u [nT][3C]
        hipLaunchKernelGGL(k_snac_local_attn, dim3(T / c.attn_window, g.C / 64, n), dim3(256), 0, st, u, y, T, g.C, c.attn_window);
        VOX_TRY(conv_gemm(st, w.attn_out, y, nullptr, nullptr, n, T, 0, off1, pk, x, nullptr, 0));                 // pk = x + out
        float* t = x; x = pk; pk = t;
    }
    size_t noise_off = 0;
    for (int bi = 0; bi < m->n_stages; ++bi) {
        const vox_snac_block_w& b = w.blocks[bi];
        const int r = c.rates[bi], cout = g.C / 2, Tin = g.Lv;
        // Snake -> compact rows + zero row; transposed conv as the 2-tap GEMM
        hipLaunchKernelGGL(k_snac_snake_pack, dim3(ew_grid((size_t)n * (Tin + 1) * g.C)), dim3(256), 0, st, x, b.snake0.alpha, b.snake0.inv_beta, pk, g);
        VOX_TRY(conv_gemm(st, b.tconv, pk, nullptr, nullptr, n, Tin + 1, 0, off2, x, nullptr, nullptr, 0));
        g = SnacGeo{n, (Tin + 1) * r, r / 2, Tin * r, cout};
        if (b.noise.n_taps > 0) {
            hipLaunchKernelGGL(k_snac_noise, dim3(ew_grid((size_t)n * g.Lb)), dim3(256), 0, st, noise ? noise + noise_off : nullptr, seed, stream_base, bi,
                               m->n_stages, m->rs, g);
            VOX_TRY(conv_gemm(st, b.noise, x, nullptr, nullptr, n, g.Lb, 0, off1, y, x, nullptr, 0, nullptr, nullptr, 0, m->rs));
            float* t = x; x = y; y = t;
            noise_off += (size_t)n * g.Lv;
        }
        static const int dils[3] = {1, 3, 9};
        for (int k = 0; k < 3; ++k) {
            const vox_snac_res_w& ru = b.res[k];
            if (ru.dense.n_taps > 0) {      // dense k7 conv: masked Snake -> 7-tap GEMM (taps read zeros outside the valid rows) with act2 fused
                const int d = dils[k];
                const int offd[14] = {3 * d, 2 * d, d, 0, -d, -2 * d, -3 * d, 3 * d, 2 * d, d, 0, -d, -2 * d, -3 * d};
                hipLaunchKernelGGL(k_snac_snake_mask, dim3(ew_grid((size_t)n * g.Lb * g.C)), dim3(256), 0, st, x, ru.act1.alpha, ru.act1.inv_beta, pk, g);
                VOX_TRY(conv_gemm(st, ru.dense, pk, nullptr, nullptr, n, g.Lb, 0, offd, nullptr, nullptr, nullptr, 0, u, &ru.act2, g.C));
            } else {
                hipLaunchKernelGGL(k_snac_dw, dim3(ew_grid((size_t)n * g.Lb * g.C)), dim3(256), 0, st, x, ru.dw_w, ru.dw_b, ru.act1.alpha, ru.act1.inv_beta,
                                   ru.act2.alpha, ru.act2.inv_beta, u, g, dils[k]);
            }
            VOX_TRY(conv_gemm(st, ru.pw, u, nullptr, nullptr, n, g.Lb, 0, off1, y, x, nullptr, 0));
            float* t = x; x = y; y = t;
        }
    }
    const int total_len = g.Lv;
    if (out_off < 0 || out_len < 1 || out_off + out_len > total_len) return vox_fail(VOX_ERR_INVALID, "snac_decode: output window out of range");
    hipLaunchKernelGGL(k_snac_final, dim3(((size_t)n * out_len + 3) / 4), dim3(256), 0, st, x, w.final_snake.alpha, w.final_snake.inv_beta, w.final_w,
                       w.final_b, out, g, out_off, out_len);
    return VOX_OK;
}

}  // extern "C"

// ====================================================================================================================
// HiFT vocoder (mel -> waveform; CosyVoice2 / GLM-4-Voice detokenizers).  See include/voxhip.h for the contract.
// Layout: fp32 time-major [request][t][C].  Every conv / transposed conv runs on conv_gemm (look-ahead taps, two weight planes);
// the harmonic source, the 16-point STFT / iSTFT and the strided source convs (Cin = 18) are direct kernels.
// ====================================================================================================================
// mel [n][Cm][T] -> x [n*T][Cp] (channels >= Cm zero)
__global__ __launch_bounds__(256) void k_hift_mel_in(const float* mel, float* x, int n, int Cm, int Cp, int T) {
    const size_t total = (size_t)n * T * Cp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const size_t row = i / Cp;
        const int b = (int)(row / T), t = (int)(row % T);
        x[i] = c < Cm ? mel[((size_t)b * Cm + c) * T + t] : 0.0f;
    }
}
// f0[row] = |w . x[row] + b|: one wave per row (hifigan.py:423-426)
__global__ __launch_bounds__(256) void k_hift_f0cls(const float* x, const float* w, float bias, float* f0, int rows, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s = fmaf(x[(size_t)row * C + c], w[c], s);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) f0[row] = fabsf(s + bias);
}
// phase at the frame rate (hifigan.py:296-313): rad = (f0 h / sr) mod 1 (what the 1/scale linear resampling of the sample-rate
// track returns exactly), cumulative sum over the frames (accumulated in double like torch.cumsum on the CPU), * 2 * pi * scale
__global__ void k_hift_phase(const float* f0, float* ph, int n, int T, int H1, float sr, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * H1) return;
    const int b = i / H1, h = i % H1;
    double cum = 0.0;
    for (int t = 0; t < T; ++t) {
        const float fn = f0[(size_t)b * T + t] * (float)(h + 1);
        float r = fn / sr;
        r = r - floorf(r);
        cum += (double)r;
        ph[((size_t)b * T + t) * H1 + h] = (((float)cum * 2.0f) * 3.14159274101257324f) * scale;
    }
}
// merged harmonic source s[b][l] (hifigan.py:314-316, 334-343, 386-388): linear interpolation of the frame-rate phase back to the
// sample rate (torch upsample_linear1d, align_corners = False), sin, voiced / unvoiced noise, tanh(linear)
__global__ __launch_bounds__(256) void k_hift_source(const float* f0, const float* ph, const float* noise, uint64_t seed,
                                                      const uint32_t* stream_base, const float* lw, float lb, float* s, int n, int T,
                                                      int H1, int scale, float rscale, float alpha, float sigma, float vth) {
    const size_t L = (size_t)T * scale, total = (size_t)n * L;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / L);
        const size_t l = i % L;
        const float uv = f0[(size_t)b * T + l / scale] > vth ? 1.0f : 0.0f;
        const float amp = uv * sigma + (1.0f - uv) * alpha / 3.0f;
        float src = rscale * ((float)l + 0.5f) - 0.5f;
        src = src < 0.0f ? 0.0f : src;
        const int i0 = (int)src, i1 = i0 + (i0 < T - 1 ? 1 : 0);
        float l1 = src - (float)i0;
        l1 = l1 < 0.0f ? 0.0f : (l1 > 1.0f ? 1.0f : l1);
        const float l0 = 1.0f - l1;
        const uint32_t stream = stream_base ? stream_base[b] + 1u : (uint32_t)(2 * b + 1);
        float acc = 0.0f;
        for (int h = 0; h < H1; ++h) {
            const float p = l0 * ph[((size_t)b * T + i0) * H1 + h] + l1 * ph[((size_t)b * T + i1) * H1 + h];
            float nz;
            if (noise) nz = noise[i * H1 + h];
            else {
                uint32_t w0, w1;
                philox4((uint32_t)(l * H1 + h), stream, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), &w0, &w1);
                const float u1 = ((float)(w0 >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);
                nz = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
            }
            const float sw = (sinf(p) * alpha) * uv + amp * nz;
            acc = fmaf(sw, lw[h], acc);
        }
        s[i] = tanhf(acc + lb);
    }
}
// SineGen v1 (GLM-4-Voice's vocoder, tokenizer/glm.py:2298-2316): theta[b][l][h] = 2 pi ((cumulative sum over samples of f0 (h+1) / sr) mod 1),
// the sum accumulated in double like torch.cumsum on the CPU.  f0 is constant over the `scale` samples of a frame, so the running sum at
// sample q of frame t is start[t] + (q + 1) F_t with F_t = fl(fl(f0 (h+1)) / sr): the per-sample additions of the serial form are exact in
// double whenever F's last bit is above the sum's ulp (24-bit F >= 2^-18 against sums < 2^14: every voiced frame), so the closed form has
// the serial sum's bits; this kernel walks the T frames per (request, harmonic) — 172 dependent additions instead of 44 032 — and
// k_hift_source_v1 evaluates the phase of each sample from start[t].
__global__ void k_hift_theta_v1(const float* f0, double* start, int n, int T, int H1, int scale, float sr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * H1) return;
    const int b = i / H1, h = i % H1;
    double cum = 0.0;
    for (int t = 0; t < T; ++t) {
        const float F = (f0[(size_t)b * T + t] * (float)(h + 1)) / sr;
        start[((size_t)b * H1 + h) * T + t] = cum;
        cum += (double)F * (double)scale;
    }
}
// merged source for SineGen v1: sin(theta + initial phase), voiced / unvoiced noise, tanh(linear).  Initial phases: -pi + 2 pi u with u
// given (rand_ini [n][H1], column 0 ignored) or word 0 of the seeded uniform stream stream_base[b] (element h); phase of the fundamental = 0
__global__ __launch_bounds__(256) void k_hift_source_v1(const float* f0, const double* start, float sr, const float* rand_ini, const float* noise, uint64_t seed,
                                                         const uint32_t* stream_base, const float* lw, float lb, float* s, int n, int T, int H1,
                                                         int scale, float alpha, float sigma, float vth) {
    const size_t L = (size_t)T * scale, total = (size_t)n * L;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / L);
        const size_t l = i % L;
        const int tf = (int)(l / scale), qf = (int)(l - (size_t)tf * scale);
        const float f0v = f0[(size_t)b * T + tf];
        const float uv = f0v > vth ? 1.0f : 0.0f;
        const float amp = uv * sigma + (1.0f - uv) * alpha / 3.0f;
        const uint32_t sb = stream_base ? stream_base[b] : (uint32_t)(2 * b);
        float acc = 0.0f;
        for (int h = 0; h < H1; ++h) {
            const float F = (f0v * (float)(h + 1)) / sr;
            const float cv = (float)(start[((size_t)b * H1 + h) * T + tf] + (double)F * (double)(qf + 1));
            const float theta = 6.28318548202514648f * (cv - floorf(cv));
            float ph = 0.0f;
            if (h > 0) {
                float u;
                if (rand_ini) u = rand_ini[(size_t)b * H1 + h];
                else {
                    uint32_t w0, w1;
                    philox4((uint32_t)h, sb, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), &w0, &w1);
                    u = (float)(w0 >> 8) * (1.0f / 16777216.0f);
                }
                ph = -3.14159274101257324f + 6.28318548202514648f * u;
            }
            float nz;
            if (noise) nz = noise[i * H1 + h];
            else {
                uint32_t w0, w1;
                philox4((uint32_t)(l * H1 + h), sb + 1u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), &w0, &w1);
                const float u1 = ((float)(w0 >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);
                nz = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
            }
            const float sw = (alpha * sinf(theta + ph)) * uv + amp * nz;
            acc = fmaf(sw, lw[h], acc);
        }
        s[i] = tanhf(acc + lb);
    }
}
// STFT of the source (torch.stft: n_fft 16, hop 4, periodic Hann, center = True with reflect padding): S[b][f][k] real rows
// 0..nb-1, imaginary rows nb..2nb-1; one thread per (b, frame, bin)
__global__ __launch_bounds__(256) void k_hift_stft(const float* s, float* S, int n, int L, int F, int nfft, int hop) {
    const int nb = nfft / 2 + 1;
    const size_t total = (size_t)n * F * nb;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % nb);
        const size_t bf = i / nb;
        const int b = (int)(bf / F), f = (int)(bf % F);
        float re = 0.0f, im = 0.0f;
        for (int j = 0; j < nfft; ++j) {
            int q = f * hop + j - nfft / 2;
            q = q < 0 ? -q : (q >= L ? 2 * (L - 1) - q : q);
            const float win = 0.5f - 0.5f * cospif(2.0f * (float)j / (float)nfft);
            const float v = s[(size_t)b * L + q] * win;
            const float ang = 2.0f * (float)((k * j) % nfft) / (float)nfft;      // in units of pi
            re = fmaf(v, cospif(ang), re);
            im = fmaf(-v, sinpif(ang), im);
        }
        S[bf * (2 * nb) + k] = re;
        S[bf * (2 * nb) + nb + k] = im;
    }
}
// source_downs[i]: strided conv over the STFT frames, Cin = 2 nb (hifigan.py:497-510): out[b][t][co]
__global__ __launch_bounds__(256) void k_hift_sd(const float* S, const float* wt, const float* bias, float* out, int n, int F, int Cin,
                                                  int Lo, int Cout, int k, int stride, int pad) {
    // wt = the weight transposed to [k][Cin][Cout] at creation: consecutive threads (output channels of one row) read consecutive
    // floats, the spectrum value is the same address for the whole row (the [Cout][Cin][k] original had every lane 4 k Cin bytes apart);
    // the accumulation order (tap, then input channel) is the original's
    const size_t total = (size_t)n * Lo * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        const size_t row = i / Cout;
        const int b = (int)(row / Lo), t = (int)(row % Lo);
        float acc = bias[co];
        for (int j = 0; j < k; ++j) {
            const int f = t * stride - pad + j;
            if (f < 0 || f >= F) continue;
            const float* sr = S + ((size_t)b * F + f) * Cin;
            const float* wr = wt + (size_t)j * Cin * Cout + co;
#pragma unroll 6
            for (int ci = 0; ci < Cin; ++ci) acc = fmaf(sr[ci], wr[(size_t)ci * Cout], acc);
        }
        out[i] = acc;
    }
}
// [Cout][Cin][k] -> [k][Cin][Cout]
__global__ __launch_bounds__(256) void k_hift_sd_transpose(const float* w, float* wt, int Cout, int Cin, int k) {
    const size_t total = (size_t)Cout * Cin * k;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % Cout), ci = (int)((i / Cout) % Cin), j = (int)(i / ((size_t)Cout * Cin));
        wt[i] = w[((size_t)co * Cin + ci) * k + j];
    }
}
// x = pad(y) + si (ReflectionPad1d((1, 0)) on the last stage: row 0 = row 1 of y) and the first Snake of the three resblocks
__global__ __launch_bounds__(256) void k_hift_add_snake3(const float* y, const float* si, float* x, float* a0, float* a1, float* a2,
                                                          const float* al0, const float* ib0, const float* al1, const float* ib1,
                                                          const float* al2, const float* ib2, int n, int Ly, int Lx, int C) {
    const size_t total = (size_t)n * Lx * C;
    const int padl = Lx - Ly;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t row = i / C;
        const int b = (int)(row / Lx), t = (int)(row % Lx);
        int ty = t - padl;
        ty = ty < 0 ? -ty : ty;
        const float v = y[((size_t)b * Ly + ty) * C + c] + si[i];
        x[i] = v;
        float s_;
        s_ = sinf(v * al0[c]); a0[i] = v + ib0[c] * (s_ * s_);
        if (a1) { s_ = sinf(v * al1[c]); a1[i] = v + ib1[c] * (s_ * s_); }
        if (a2) { s_ = sinf(v * al2[c]); a2[i] = v + ib2[c] * (s_ * s_); }
    }
}
// x = leaky_relu(((r0 + r1) + r2) / nk, slope)   (hifigan.py:613-621; r1 / r2 NULL when fewer kernels)
__global__ __launch_bounds__(256) void k_hift_avg_lrelu(const float* r0, const float* r1, const float* r2, float* x, size_t total, float nk,
                                                         float slope) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float v = r0[i];
        if (r1) v = v + r1[i];
        if (r2) v = v + r2[i];
        if (nk != 1.0f) v = v / nk;
        x[i] = v > 0.0f ? v : v * slope;
    }
}
// magnitude / phase heads -> irfft per frame -> window -> overlap-add / window^2 -> trim n_fft/2 -> clamp  (hifigan.py:554-594, 623-628)
// one thread per output sample; P[b][f][0..nb-1] = log-magnitude rows, [nb..2nb-1] = phase rows
__global__ __launch_bounds__(256) void k_hift_istft(const float* P, float* wav, int n, int F, int nfft, int hop, int Lout, float limit) {
    const int nb = nfft / 2 + 1;
    const size_t total = (size_t)n * Lout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / Lout), l = (int)(i % Lout);
        const int q = l + nfft / 2;                        // position in the untrimmed signal
        float acc = 0.0f, den = 0.0f;
        int f_hi = q / hop;
        f_hi = f_hi > F - 1 ? F - 1 : f_hi;
        int f_lo = (q - nfft) / hop + 1;
        f_lo = (q - nfft) < 0 ? 0 : f_lo;
        for (int f = f_lo; f <= f_hi; ++f) {               // ascending frame order, like F.fold's accumulation
            const int j = q - f * hop;
            if (j < 0 || j >= nfft) continue;
            const float* pf = P + ((size_t)b * F + f) * (2 * nb);
            float v = 0.0f;
            for (int k = 0; k < nb; ++k) {
                float mag = expf(pf[k]);
                mag = mag > 1e2f ? 1e2f : mag;
                const float ph = sinf(pf[nb + k]);
                const float re = mag * cosf(ph), im = mag * sinf(ph);
                const float ang = 2.0f * (float)((k * j) % nfft) / (float)nfft;
                if (k == 0 || k == nb - 1) v += re * cospif(ang);                       // irfft ignores the imaginary part of DC / Nyquist
                else v += 2.0f * (re * cospif(ang) - im * sinpif(ang));
            }
            const float win = 0.5f - 0.5f * cospif(2.0f * (float)j / (float)nfft);
            acc += (v / (float)nfft) * win;
            den += win * win;
        }
        den = den > 1e-8f ? den : 1.0f;
        float o = acc / den;
        o = o < -limit ? -limit : (o > limit ? limit : o);
        wav[i] = o;
    }
}

struct vox_hift {
    vox_ctx* ctx;
    vox_hift_config cfg;
    vox_hift_weights w;
    int max_batch, max_T, scale;
    float* buf[9];
    size_t buf_floats;
    float *f0, *ph, *src, *stft, *post;
    double* theta = nullptr;               // SineGen v1: running phase sum at the start of every frame [n][H1][T]
    float* sd_wt[4] = {};                  // source_downs weights as [k][n_fft + 2][Cout] (k_hift_sd)
    int sine_v1 = 0;
};

static int hift_conv_offsets(int k, int dil, int* off, int planes = 2) {   // "same" conv: tap j reads row t + (j - (k-1)/2) dil
    const int pad = (k - 1) * dil / 2;
    for (int p = 0; p < planes; ++p)
        for (int j = 0; j < k; ++j) off[p * k + j] = pad - j * dil;
    return planes * k;
}
static int hift_tconv_offsets(int k, int u, int* off) {           // ConvTranspose1d(k, stride u, padding (k-u)/2): taps d = dmin..dmax, two planes
    const int p = (k - u) / 2;
    int dmin = -((p + (u - 1)) / u), dmax = (k - 1 - p) / u;
    while (dmin * u + (u - 1) + p < 0) ++dmin;
    const int nd = dmax - dmin + 1;
    for (int pl = 0; pl < 2; ++pl)
        for (int d = 0; d < nd; ++d) off[pl * nd + d] = dmin + d;
    return 2 * nd;
}

extern "C" {

void vox_hift_destroy(vox_hift* m) {
    if (!m) return;
    for (int i = 0; i < 9; ++i) (void)hipFree(m->buf[i]);
    (void)hipFree(m->f0); (void)hipFree(m->ph); (void)hipFree(m->src); (void)hipFree(m->stft); (void)hipFree(m->post); (void)hipFree(m->theta);
    for (int i = 0; i < 4; ++i) (void)hipFree(m->sd_wt[i]);
    delete m;
}

int vox_hift_create(vox_ctx* ctx, const vox_hift_config* cfg, const vox_hift_weights* w, int max_batch, int max_T, vox_hift** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "hift_create: NULL");
    if (cfg->n_stages < 1 || cfg->n_stages > 4 || cfg->n_kernels < 1 || cfg->n_kernels > 3 || max_batch < 1 || max_T < 1 ||
        cfg->in_channels_padded % 32 || cfg->in_channels > cfg->in_channels_padded || cfg->n_fft % 2 || cfg->nb_harmonics > 15)
        return vox_fail(VOX_ERR_INVALID, "hift_create: bad config");
    vox_hift* m = new vox_hift();
    m->ctx = ctx; m->cfg = *cfg; m->w = *w; m->max_batch = max_batch; m->max_T = max_T;
    m->scale = cfg->hop_len;
    size_t rows = max_T, worst = (size_t)max_T * cfg->base_channels;
    int ch = cfg->base_channels;
    if ((size_t)max_T * cfg->f0_channels > worst) worst = (size_t)max_T * cfg->f0_channels;
    if ((size_t)max_T * cfg->in_channels_padded > worst) worst = (size_t)max_T * cfg->in_channels_padded;
    for (int i = 0; i < cfg->n_stages; ++i) {
        if ((cfg->upsample_kernels[i] - cfg->upsample_rates[i]) < 0) { delete m; return vox_fail(VOX_ERR_INVALID, "hift_create: kernel < stride"); }
        rows *= cfg->upsample_rates[i]; ch /= 2; m->scale *= cfg->upsample_rates[i];
        if ((rows + 1) * ch > worst) worst = (rows + 1) * ch;
        if (ch % 32) { delete m; return vox_fail(VOX_ERR_INVALID, "hift_create: stage channels must be multiples of 32"); }
    }
    m->buf_floats = worst * max_batch;
    const size_t L = (size_t)max_T * m->scale, F = rows + 1;
    bool ok = true;
    for (int i = 0; i < 9; ++i) ok = ok && hipMalloc((void**)&m->buf[i], m->buf_floats * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&m->f0, (size_t)max_batch * max_T * 4) == hipSuccess &&
         hipMalloc((void**)&m->ph, (size_t)max_batch * max_T * (cfg->nb_harmonics + 1) * 4) == hipSuccess &&
         hipMalloc((void**)&m->src, (size_t)max_batch * L * 4) == hipSuccess &&
         hipMalloc((void**)&m->stft, (size_t)max_batch * F * (cfg->n_fft + 2) * 4) == hipSuccess &&
         hipMalloc((void**)&m->post, (size_t)max_batch * F * (cfg->n_fft + 2) * 4) == hipSuccess;
    m->sine_v1 = cfg->sine_gen_v1;
    if (m->sine_v1) ok = ok && hipMalloc((void**)&m->theta, (size_t)max_batch * max_T * (cfg->nb_harmonics + 1) * 8) == hipSuccess;
    ch = cfg->base_channels;
    for (int i = 0; i < cfg->n_stages && ok; ++i) {      // transposed source_downs weights
        int sstride = 1;
        for (int q = i + 1; q < cfg->n_stages; ++q) sstride *= cfg->upsample_rates[q];
        const int sk = sstride == 1 ? 1 : 2 * sstride, cin = cfg->n_fft + 2;
        ch /= 2;
        const size_t cnt = (size_t)ch * cin * sk;
        ok = hipMalloc((void**)&m->sd_wt[i], cnt * 4) == hipSuccess;
        if (ok) hipLaunchKernelGGL(k_hift_sd_transpose, dim3(ew_grid(cnt)), dim3(256), 0, 0, w->sd_w[i], m->sd_wt[i], ch, cin, sk);
    }
    ok = ok && hipStreamSynchronize(0) == hipSuccess;
    if (!ok) { vox_hift_destroy(m); return vox_fail(VOX_ERR_NOMEM, "hift_create: hipMalloc failed"); }
    *out = m;
    return VOX_OK;
}

// one ResBlock (hifigan.py:134-141) on rows L per request: a = snake1_0(x) is given; result in `r` (x itself is left untouched);
// t1 scratch.  The next iteration's first Snake and this iteration's second Snake are fused into the producing conv's epilogue.
static int hift_resblock(hipStream_t st, const vox_hift_resblock_w& rb, const vox_hift_config& c, int k, const float* x, float* a, float* t1,
                         float* r, int n, int L) {
    int off1[CG_MAXTAPS], off2[CG_MAXTAPS];
    hift_conv_offsets(k, 1, off2);
    for (int j = 0; j < 3; ++j) {
        hift_conv_offsets(k, c.dilations[j], off1);
        VOX_TRY(conv_gemm(st, rb.c1[j], a, nullptr, nullptr, n, L, 0, off1, nullptr, nullptr, nullptr, 0, t1, &rb.a2[j]));
        VOX_TRY(conv_gemm(st, rb.c2[j], t1, nullptr, nullptr, n, L, 0, off2, r, j == 0 ? x : r, nullptr, 0, j < 2 ? a : nullptr,
                          j < 2 ? &rb.a1[j + 1] : nullptr));
    }
    return VOX_OK;
}

int vox_hift_decode(vox_hift* m, void* stream, const float* mel, int n, int T, const float* noise, uint64_t seed,
                    const uint32_t* stream_base, float* wav, float* source, const float* rand_ini) {
    if (!m || !mel || !wav) return vox_fail(VOX_ERR_INVALID, "hift_decode: NULL");
    if (n < 1 || n > m->max_batch || T < 1 || T > m->max_T) return vox_fail(VOX_ERR_INVALID, "hift_decode: n %d / T %d out of range", n, T);
    const vox_hift_config& c = m->cfg;
    const vox_hift_weights& w = m->w;
    hipStream_t st = (hipStream_t)stream;
    g_conv_planes = 3;
    g_conv_skinny_rows = 48; g_conv_rows_gemm = false;
    const int H1 = c.nb_harmonics + 1, nb2 = c.n_fft + 2;
    const size_t L = (size_t)T * m->scale;
    float** B = m->buf;
    int off[CG_MAXTAPS];
    // ---- f0 predictor: 5 x (conv k3 + ELU), linear, abs ----
    hipLaunchKernelGGL(k_hift_mel_in, dim3(ew_grid((size_t)n * T * c.in_channels_padded)), dim3(256), 0, st, mel, B[0], n, c.in_channels,
                       c.in_channels_padded, T);
    hift_conv_offsets(3, 1, off, 3);          // (three weight planes: the source's phase multiplies an f0 error by 2 pi scale T)
    {
        const float* in = B[0];
        for (int i = 0; i < 5; ++i) {
            float* o = (i & 1) ? B[2] : B[1];
            VOX_TRY(conv_gemm(st, w.f0_conv[i], in, nullptr, nullptr, n, T, 0, off, o, nullptr, nullptr, 0));
            elu(st, o, o, (size_t)n * T * c.f0_channels);
            in = o;
        }
        hipLaunchKernelGGL(k_hift_f0cls, dim3((n * T + 3) / 4), dim3(256), 0, st, in, w.f0_cls_w, w.f0_cls_b, m->f0, n * T, c.f0_channels);
    }
    // ---- harmonic source + its STFT ----
    if (m->sine_v1) {
        hipLaunchKernelGGL(k_hift_theta_v1, dim3((n * H1 + 63) / 64), dim3(64), 0, st, m->f0, m->theta, n, T, H1, m->scale, (float)c.sampling_rate);
        hipLaunchKernelGGL(k_hift_source_v1, dim3(ew_grid((size_t)n * L)), dim3(256), 0, st, m->f0, m->theta, (float)c.sampling_rate, rand_ini, noise, seed, stream_base,
                           w.src_lin_w, w.src_lin_b, m->src, n, T, H1, m->scale, c.nsf_alpha, c.nsf_sigma, c.voiced_threshold);
    } else {
        hipLaunchKernelGGL(k_hift_phase, dim3((n * H1 + 63) / 64), dim3(64), 0, st, m->f0, m->ph, n, T, H1, (float)c.sampling_rate, (float)m->scale);
        hipLaunchKernelGGL(k_hift_source, dim3(ew_grid((size_t)n * L)), dim3(256), 0, st, m->f0, m->ph, noise, seed, stream_base, w.src_lin_w,
                           w.src_lin_b, m->src, n, T, H1, m->scale, (float)(1.0 / (double)m->scale), c.nsf_alpha, c.nsf_sigma, c.voiced_threshold);
    }
    if (source) (void)hipMemcpyAsync(source, m->src, (size_t)n * L * 4, hipMemcpyDeviceToDevice, st);
    const int F = (int)(L / c.hop_len) + 1;
    hipLaunchKernelGGL(k_hift_stft, dim3(ew_grid((size_t)n * F * (c.n_fft / 2 + 1))), dim3(256), 0, st, m->src, m->stft, n, (int)L, F, c.n_fft,
                       c.hop_len);
    // ---- conv_pre ----
    hift_conv_offsets(7, 1, off);
    VOX_TRY(conv_gemm(st, w.conv_pre, B[0], nullptr, nullptr, n, T, 0, off, B[1], nullptr, nullptr, 0));
    hipLaunchKernelGGL(k_hift_avg_lrelu, dim3(ew_grid((size_t)n * T * c.base_channels)), dim3(256), 0, st, B[1], nullptr, nullptr, B[1],
                       (size_t)n * T * c.base_channels, 1.0f, c.lrelu_slope);
    float* cur = B[1];                     // leaky_relu'ed stage input
    int Lc = T, ch = c.base_channels;
    // strides of source_downs: cumulative products of the LATER stages' rates (hifigan.py:497-500)
    for (int i = 0; i < c.n_stages; ++i) {
        const int u = c.upsample_rates[i], k = c.upsample_kernels[i], cout = ch / 2;
        const bool last = i == c.n_stages - 1;
        const int Ly = Lc * u, Lx = last ? Ly + 1 : Ly;
        // pick working buffers distinct from `cur`
        float* pool[8];
        int np = 0;
        for (int q = 0; q < 9; ++q) if (B[q] != cur) pool[np++] = B[q];
        float *Y = pool[0], *S = pool[1], *A = pool[2], *T1 = pool[3], *X = pool[4], *A1 = pool[5], *A2 = pool[6], *R2 = pool[7];
        hift_tconv_offsets(k, u, off);
        VOX_TRY(conv_gemm(st, w.ups[i], cur, nullptr, nullptr, n, Lc, 0, off, Y, nullptr, nullptr, 0));
        int sstride = 1;
        for (int q = i + 1; q < c.n_stages; ++q) sstride *= c.upsample_rates[q];
        const int sk = sstride == 1 ? 1 : 2 * sstride, spad = sstride == 1 ? 0 : sstride / 2;
        if ((F + 2 * spad - sk) / sstride + 1 != Lx) return vox_fail(VOX_ERR_INVALID, "hift_decode: source / stage length mismatch at stage %d", i);
        hipLaunchKernelGGL(k_hift_sd, dim3(ew_grid((size_t)n * Lx * cout)), dim3(256), 0, st, m->stft, m->sd_wt[i], w.sd_b[i], S, n, F, nb2, Lx, cout,
                           sk, sstride, spad);
        // source resblock: S <- resblock(S)
        snake(st, S, w.src_rb[i].a1[0], A, (size_t)n * Lx, cout);
        VOX_TRY(hift_resblock(st, w.src_rb[i], c, c.source_resblock_kernels[i], S, A, T1, X, n, Lx));       // result in X (scratch here)
        // x = pad(Y) + si, first Snakes of the stage's resblocks
        const vox_hift_resblock_w* rb = &w.rb[i * c.n_kernels];
        float* Si = X;                      // source branch output
        float* Xs = S;                      // stage tensor (S is free now)
        hipLaunchKernelGGL(k_hift_add_snake3, dim3(ew_grid((size_t)n * Lx * cout)), dim3(256), 0, st, Y, Si, Xs, A, c.n_kernels > 1 ? A1 : nullptr,
                           c.n_kernels > 2 ? A2 : nullptr, rb[0].a1[0].alpha, rb[0].a1[0].inv_beta, c.n_kernels > 1 ? rb[1].a1[0].alpha : nullptr,
                           c.n_kernels > 1 ? rb[1].a1[0].inv_beta : nullptr, c.n_kernels > 2 ? rb[2].a1[0].alpha : nullptr,
                           c.n_kernels > 2 ? rb[2].a1[0].inv_beta : nullptr, n, Ly, Lx, cout);
        float* Rs[3] = {Y, Si, R2};         // Y and the source output are free after the add
        float* As[3] = {A, A1, A2};
        for (int r = 0; r < c.n_kernels; ++r) VOX_TRY(hift_resblock(st, rb[r], c, c.resblock_kernels[r], Xs, As[r], T1, Rs[r], n, Lx));
        // average + the next stage's leaky_relu (slope 0.01 before conv_post: F.leaky_relu's default)
        hipLaunchKernelGGL(k_hift_avg_lrelu, dim3(ew_grid((size_t)n * Lx * cout)), dim3(256), 0, st, Rs[0], c.n_kernels > 1 ? Rs[1] : nullptr,
                           c.n_kernels > 2 ? Rs[2] : nullptr, Xs, (size_t)n * Lx * cout, (float)c.n_kernels, last ? 0.01f : c.lrelu_slope);
        cur = Xs; Lc = Lx; ch = cout;
    }
    if (Lc != F) return vox_fail(VOX_ERR_INVALID, "hift_decode: frame count mismatch");
    hift_conv_offsets(7, 1, off);
    VOX_TRY(conv_gemm(st, w.conv_post, cur, nullptr, nullptr, n, Lc, 0, off, m->post, nullptr, nullptr, 0));
    hipLaunchKernelGGL(k_hift_istft, dim3(ew_grid((size_t)n * L)), dim3(256), 0, st, m->post, wav, n, F, c.n_fft, c.hop_len, (int)L, c.audio_limit);
    return VOX_OK;
}

}  // extern "C"

// ====================================================================================================================
// CosyVoice2 flow (speech tokens -> mel): conformer encoder + conditional flow matching.  See include/voxhip.h for the contract.
// ====================================================================================================================
__device__ __forceinline__ float mish_f(float v) { return v * tanhf(v > 20.0f ? v : log1pf(expf(v))); }

// x[row] = embedding[max(id, 0)]
__global__ __launch_bounds__(256) void k_flow_embed(const int* ids, const float* emb, float* x, size_t rows, int D) {
    const size_t total = rows * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int id = ids[i / D];
        x[i] = emb[(size_t)(id < 0 ? 0 : id) * D + i % D];
    }
}
// y = act(LayerNorm(x) * w + b) * post + add[row / rows_per_req]      (act 0 none, 1 Mish)
__global__ __launch_bounds__(256) void k_flow_ln(const float* x, const float* w, const float* b, float* y, int C, float eps, float post, int act,
                                                  const float* add, int rows_per_req, int rows, int ldx = 0, int ld_add = 0) {
    // ldx / ld_add: row strides of x and of add (0 = C): the ResNet blocks' first conv carries the residual conv as extra columns
    // one wave per row (4 rows per block): each of the wave's four 16-lane groups computes the row statistics (ln_row_stats: the bits the
    // fused-LayerNorm GEMMs get), then the 64 lanes apply them to one float4 chunk each per pass — a 112-row call is 28 blocks with 4 Mish
    // evaluations per lane instead of 7 blocks with 16
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * (ldx ? ldx : C);
    const float* ad = add ? add + (size_t)(row / rows_per_req) * (ld_add ? ld_add : C) : nullptr;
    if (C == 256) {
        // 256-channel rows (every LayerNorm of the CosyVoice2 estimator): lane (l16 = lane % 16, k = lane / 16) of ln_row_stats' scheme
        // owns chunks l16, l16 + 16, l16 + 32, l16 + 48 — loaded once, both statistics passes run on the registers (same order: the
        // lane's chunks ascending, then the 16-lane butterfly), and the chunk this lane writes (index `lane` = l16 + 16 k) is its k-th
        const int l16 = lane & 15, kq = lane >> 4;
        const float4* x4 = reinterpret_cast<const float4*>(xr);
        const float4 c0 = x4[l16], c1 = x4[l16 + 16], c2 = x4[l16 + 32], c3 = x4[l16 + 48];
        const float4 wv = reinterpret_cast<const float4*>(w)[lane], bv = reinterpret_cast<const float4*>(b)[lane];
        float sm = 0.0f;
        sm += (c0.x + c0.y) + (c0.z + c0.w); sm += (c1.x + c1.y) + (c1.z + c1.w);
        sm += (c2.x + c2.y) + (c2.z + c2.w); sm += (c3.x + c3.y) + (c3.z + c3.w);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) sm += __shfl_xor(sm, off, 64);
        const float mean = sm / 256.0f;
        float vs = 0.0f;
        {
            float d0 = c0.x - mean, d1 = c0.y - mean, d2 = c0.z - mean, d3 = c0.w - mean;
            vs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            d0 = c1.x - mean; d1 = c1.y - mean; d2 = c1.z - mean; d3 = c1.w - mean;
            vs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            d0 = c2.x - mean; d1 = c2.y - mean; d2 = c2.z - mean; d3 = c2.w - mean;
            vs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            d0 = c3.x - mean; d1 = c3.y - mean; d2 = c3.z - mean; d3 = c3.w - mean;
            vs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) vs += __shfl_xor(vs, off, 64);
        const float rstd = rsqrtf(vs / 256.0f + eps);
        const float4 xv = kq == 0 ? c0 : kq == 1 ? c1 : kq == 2 ? c2 : c3;
        float o[4] = {(xv.x - mean) * rstd * wv.x + bv.x, (xv.y - mean) * rstd * wv.y + bv.y, (xv.z - mean) * rstd * wv.z + bv.z, (xv.w - mean) * rstd * wv.w + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (act == 1) o[e] = mish_f(o[e]);
            o[e] *= post;
        }
        if (ad) { const float4 av = reinterpret_cast<const float4*>(ad)[lane]; o[0] += av.x; o[1] += av.y; o[2] += av.z; o[3] += av.w; }
        reinterpret_cast<float4*>(y + (size_t)row * C)[lane] = make_float4(o[0], o[1], o[2], o[3]);
        return;
    }
    float mean, rstd;
    ln_row_stats(xr, C, lane & 15, eps, mean, rstd);
    for (int i = lane; i < C / 4; i += 64) {
        const float4 xv = reinterpret_cast<const float4*>(xr)[i], wv = reinterpret_cast<const float4*>(w)[i], bv = reinterpret_cast<const float4*>(b)[i];
        float o[4] = {(xv.x - mean) * rstd * wv.x + bv.x, (xv.y - mean) * rstd * wv.y + bv.y, (xv.z - mean) * rstd * wv.z + bv.z, (xv.w - mean) * rstd * wv.w + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (act == 1) o[e] = mish_f(o[e]);
            o[e] *= post;
        }
        if (ad) { const float4 av = reinterpret_cast<const float4*>(ad)[i]; o[0] += av.x; o[1] += av.y; o[2] += av.z; o[3] += av.w; }
        reinterpret_cast<float4*>(y + (size_t)row * C)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// in-place activation: 1 Mish, 2 SiLU, 3 leaky_relu(0.01)
__global__ __launch_bounds__(256) void k_flow_act(float* x, size_t total, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        x[i] = act == 1 ? mish_f(v) : act == 2 ? v / (1.0f + expf(-v)) : (v > 0.0f ? v : 0.01f * v);
    }
}
// nearest x2 along time: y[b][2t + j] = x[b][t]
__global__ __launch_bounds__(256) void k_flow_repeat2(const float* x, float* y, size_t rows, int C) {
    const size_t total = rows * 2 * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) y[i] = x[(i / C / 2) * C + i % C];
}
// EspnetRelPositionalEncoding.position_encoding(0, S): pe[j] = encoding of position S - 1 - j   (cosyvoice_flow.py:427-447, 483-486)
__global__ __launch_bounds__(256) void k_flow_relpos(float* pe, int S, int D) {
    const size_t total = (size_t)(2 * S - 1) * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i / D), c = (int)(i % D);
        const float pos = (float)(S - 1 - j);
        const float div = expf((float)(c & ~1) * (float)(-(9.210340371976184 / (double)D)));
        pe[i] = (c & 1) ? cosf(pos * div) : sinf(pos * div);
    }
}
// Attention over [cached keys ; this chunk's keys], no mask.  One wave per (request n, head h, query i).
// qkv [N * T][3 HD] (q | k | v); cache [chalf][H][Tcap][2 dk] (K | V), Tc valid; output [N * T][HD].
// P != NULL: the relative-position term of RelPositionMultiHeadedAttention with its literal rel_shift (cosyvoice_flow.py:772-786, 830-857).
struct FlowAttn {
    const float *qkv, *cache, *P, *bu, *bv;
    float* out;
    int T, H, dk, Tc, Tcap, B;          // B: requests per classifier-free-guidance half (cache half = n / B); B <= 0: one shared cache
    size_t cache_half_stride;
    float scale;
    int mask_block;                     // > 0: key j is visible to query i iff j <= i or j / mask_block == i / mask_block (glm.py:452-470)
    // per-request evolving caches (CosyVoice2 use_detokenizer_cache): row n reads cache block cidx[n] (stride cache_half_stride); the
    // cached keys beyond the first `prefix` rows sit in a ring that starts at ring_head (logical row j -> physical row below)
    const int* cidx;
    int prefix, ring_head;
};
__device__ __forceinline__ int flow_cache_row(int j, int prefix, int head, int cap) {
    return (head == 0 || j < prefix) ? j : prefix + (head + j - prefix) % (cap - prefix);
}
#define FLOW_MAXKEYS 1024
__global__ __launch_bounds__(256) void k_flow_attn(FlowAttn a) {
    __shared__ float ps[4][FLOW_MAXKEYS];
    __shared__ __attribute__((aligned(16))) float qs[4][2][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave, h = blockIdx.y, n = blockIdx.z;
    if (i >= a.T) return;
    const int HD = a.H * a.dk, ld = 3 * HD, S = a.Tc + a.T;
    const float* cache = a.cache ? a.cache + (a.cidx ? (size_t)a.cidx[n] * a.cache_half_stride : a.B > 0 ? (size_t)(n / a.B) * a.cache_half_stride : 0) +
                                       (size_t)h * a.Tcap * 2 * a.dk : nullptr;
    const float* qrow = a.qkv + ((size_t)n * a.T + i) * ld + h * a.dk;
    for (int d = lane; d < a.dk; d += 64) {
        qs[wave][0][d] = qrow[d] + (a.bu ? a.bu[h * a.dk + d] : 0.0f);
        qs[wave][1][d] = a.bv ? a.bv[h * a.dk + d] : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    const int dk4 = a.dk >> 2;
    const float4* q4 = reinterpret_cast<const float4*>(qs[wave][0]);
    const float4* v4 = reinterpret_cast<const float4*>(qs[wave][1]);
    for (int j = lane; j < S; j += 64) {
        const float4* kr = reinterpret_cast<const float4*>(j < a.Tc ? cache + (size_t)flow_cache_row(j, a.prefix, a.ring_head, a.Tcap) * 2 * a.dk
                                                                    : a.qkv + ((size_t)n * a.T + (j - a.Tc)) * ld + HD + h * a.dk);
        float ac = 0.0f;
#pragma unroll 4
        for (int d = 0; d < dk4; ++d) {
            const float4 k = kr[d], q = q4[d];
            ac = fmaf(q.x, k.x, ac); ac = fmaf(q.y, k.y, ac); ac = fmaf(q.z, k.z, ac); ac = fmaf(q.w, k.w, ac);
        }
        if (a.P) {
            const long f = (long)i * (2 * S - 1) + j + a.T;
            const int r = (int)(f / (2 * S)), ci = (int)(f % (2 * S));
            if (ci > 0) {
                const float4* qr = reinterpret_cast<const float4*>(a.qkv + ((size_t)n * a.T + r) * ld + h * a.dk);
                const float4* pr = reinterpret_cast<const float4*>(a.P + (size_t)(ci - 1) * HD + h * a.dk);
                float bd = 0.0f;
#pragma unroll 4
                for (int d = 0; d < dk4; ++d) {
                    const float4 q = qr[d], bv = v4[d], p = pr[d];
                    bd = fmaf(q.x + bv.x, p.x, bd); bd = fmaf(q.y + bv.y, p.y, bd); bd = fmaf(q.z + bv.z, p.z, bd); bd = fmaf(q.w + bv.w, p.w, bd);
                }
                ac += bd;
            }
        }
        ac *= a.scale;
        if (a.mask_block > 0 && j > i && j / a.mask_block != i / a.mask_block) ac = -INFINITY;
        ps[wave][j] = ac;
        mx = fmaxf(mx, ac);
    }
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.0f;
    for (int j = lane; j < S; j += 64) {
        const float p = expf(ps[wave][j] - mx);
        ps[wave][j] = p;
        sum += p;
    }
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / sum;
    // PV: lane = (key group g = lane / 16, float4 chunk c = lane % 16 of the output dims); a pass reads four whole V rows (coalesced
    // 256-byte segments), four passes are in flight together; the four key groups are summed by two butterfly steps at the end.
    // (chunk c covers dims 4c .. 4c+3 of each 64-dim slab; lanes past dk idle)
    for (int c4 = (lane & 15) * 4, slab = 0; slab < a.dk; slab += 64, c4 += 64) {
        const int g = lane >> 4;
        const bool on = c4 < a.dk;
        float4 o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* vc = cache ? cache + a.dk + c4 : nullptr;
        const float* vn = a.qkv + (size_t)n * a.T * ld + 2 * HD + h * a.dk + c4;
        auto vrow = [&](int j) {
            return reinterpret_cast<const float4*>(j < a.Tc ? vc + (size_t)flow_cache_row(j, a.prefix, a.ring_head, a.Tcap) * 2 * a.dk : vn + (size_t)(j - a.Tc) * ld);
        };
        int j = g;
        if (on) {
            for (; j + 12 < S; j += 16) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *vrow(j + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float p = ps[wave][j + 4 * u];
                    o[u].x = fmaf(p, v[u].x, o[u].x); o[u].y = fmaf(p, v[u].y, o[u].y);
                    o[u].z = fmaf(p, v[u].z, o[u].z); o[u].w = fmaf(p, v[u].w, o[u].w);
                }
            }
            for (; j < S; j += 4) {
                const float4 v = *vrow(j);
                const float p = ps[wave][j];
                o[0].x = fmaf(p, v.x, o[0].x); o[0].y = fmaf(p, v.y, o[0].y); o[0].z = fmaf(p, v.z, o[0].z); o[0].w = fmaf(p, v.w, o[0].w);
            }
        }
        float4 r = make_float4((o[0].x + o[1].x) + (o[2].x + o[3].x), (o[0].y + o[1].y) + (o[2].y + o[3].y),
                               (o[0].z + o[1].z) + (o[2].z + o[3].z), (o[0].w + o[1].w) + (o[2].w + o[3].w));
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            r.x += __shfl_xor(r.x, off, 64); r.y += __shfl_xor(r.y, off, 64); r.z += __shfl_xor(r.z, off, 64); r.w += __shfl_xor(r.w, off, 64);
        }
        if (on && g == 0)
            *reinterpret_cast<float4*>(a.out + ((size_t)n * a.T + i) * HD + h * a.dk + c4) = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
    }
}
// The same attention for the calls without a relative-position term (the CFM estimators' transformer blocks: thousands of launches per
// chunk), with the (request, head)'s keys and values staged ONCE per block in LDS: k_flow_attn has every wave stream all S key and value
// rows from L2 for its one query (94 KB per query at S = 184, dk = 64: 84 MB of L2 reads per call at one request).  A block owns QG
// consecutive queries, its four waves take them in turn; arithmetic and summation orders are k_flow_attn's, so the two kernels agree
// bit for bit (which one a call takes may depend on its batch size; a request's output may not).  K rows are padded to DK + 4 floats:
// lane = key reads of a float4 column then touch every bank once per 16 lanes.
template <int DK>
__global__ __launch_bounds__(256) void k_flow_attn_tile(FlowAttn a, int QG) {
    extern __shared__ __attribute__((aligned(16))) float fa_sm[];
    constexpr int KS = DK + 4, DK4 = DK / 4;
    const int S = a.Tc + a.T, S4 = (S + 3) & ~3;
    float* Ks = fa_sm;
    float* Vs = Ks + (size_t)S * KS;
    float* psm = Vs + (size_t)S * DK;
    float* qsm = psm + 4 * S4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int h = blockIdx.y, n = blockIdx.z, i0 = blockIdx.x * QG;
    const int HD = a.H * DK, ld = 3 * HD;
    const float* cache = a.cache ? a.cache + (a.cidx ? (size_t)a.cidx[n] * a.cache_half_stride : a.B > 0 ? (size_t)(n / a.B) * a.cache_half_stride : 0) +
                                       (size_t)h * a.Tcap * 2 * DK : nullptr;
    for (int idx = tid; idx < S * DK4; idx += 256) {
        const int j = idx / DK4, c = idx - j * DK4;
        const float *kr, *vr;
        if (j < a.Tc) {
            kr = cache + (size_t)flow_cache_row(j, a.prefix, a.ring_head, a.Tcap) * 2 * DK;
            vr = kr + DK;
        } else {
            kr = a.qkv + ((size_t)n * a.T + (j - a.Tc)) * ld + HD + h * DK;
            vr = kr + HD;
        }
        *reinterpret_cast<float4*>(Ks + (size_t)j * KS + 4 * c) = reinterpret_cast<const float4*>(kr)[c];
        *reinterpret_cast<float4*>(Vs + (size_t)j * DK + 4 * c) = reinterpret_cast<const float4*>(vr)[c];
    }
    __syncthreads();
    float* ps = psm + wave * S4;
    float* qs = qsm + wave * DK;
    const int iend = (i0 + QG < a.T) ? i0 + QG : a.T;
    for (int i = i0 + wave; i < iend; i += 4) {
        const float* qrow = a.qkv + ((size_t)n * a.T + i) * ld + h * DK;
        for (int d = lane; d < DK; d += 64) qs[d] = qrow[d] + (a.bu ? a.bu[h * DK + d] : 0.0f);
        __builtin_amdgcn_wave_barrier();
        float mx = -INFINITY;
        const float4* q4 = reinterpret_cast<const float4*>(qs);
        for (int j = lane; j < S; j += 64) {
            const float4* kr = reinterpret_cast<const float4*>(Ks + (size_t)j * KS);
            float ac = 0.0f;
#pragma unroll
            for (int d = 0; d < DK4; ++d) {
                const float4 k = kr[d], q = q4[d];
                ac = fmaf(q.x, k.x, ac); ac = fmaf(q.y, k.y, ac); ac = fmaf(q.z, k.z, ac); ac = fmaf(q.w, k.w, ac);
            }
            ac *= a.scale;
            if (a.mask_block > 0 && j > i && j / a.mask_block != i / a.mask_block) ac = -INFINITY;
            ps[j] = ac;
            mx = fmaxf(mx, ac);
        }
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float sum = 0.0f;
        for (int j = lane; j < S; j += 64) {
            const float p = expf(ps[j] - mx);
            ps[j] = p;
            sum += p;
        }
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / sum;
        // PV in k_flow_attn's order: lane = (key group g, float4 chunk c), keys g, g + 4, ... in four interleaved accumulators
        for (int c4 = (lane & 15) * 4, slab = 0; slab < DK; slab += 64, c4 += 64) {
            const int g = lane >> 4;
            const bool on = c4 < DK;
            float4 o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            int j = g;
            if (on) {
                for (; j + 12 < S; j += 16) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(Vs + (size_t)(j + 4 * u) * DK + c4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float p = ps[j + 4 * u];
                        o[u].x = fmaf(p, v[u].x, o[u].x); o[u].y = fmaf(p, v[u].y, o[u].y);
                        o[u].z = fmaf(p, v[u].z, o[u].z); o[u].w = fmaf(p, v[u].w, o[u].w);
                    }
                }
                for (; j < S; j += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(Vs + (size_t)j * DK + c4);
                    const float p = ps[j];
                    o[0].x = fmaf(p, v.x, o[0].x); o[0].y = fmaf(p, v.y, o[0].y); o[0].z = fmaf(p, v.z, o[0].z); o[0].w = fmaf(p, v.w, o[0].w);
                }
            }
            float4 r = make_float4((o[0].x + o[1].x) + (o[2].x + o[3].x), (o[0].y + o[1].y) + (o[2].y + o[3].y),
                                   (o[0].z + o[1].z) + (o[2].z + o[3].z), (o[0].w + o[1].w) + (o[2].w + o[3].w));
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                r.x += __shfl_xor(r.x, off, 64); r.y += __shfl_xor(r.y, off, 64); r.z += __shfl_xor(r.z, off, 64); r.w += __shfl_xor(r.w, off, 64);
            }
            if (on && g == 0)
                *reinterpret_cast<float4*>(a.out + ((size_t)n * a.T + i) * HD + h * DK + c4) = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
        }
        __builtin_amdgcn_wave_barrier();      // ps / qs are rewritten by the wave's next query
    }
}
// The estimators' attention on the matrix cores (no relative-position term, dk = 64, at most 256 keys: every CosyVoice2 / GLM estimator
// call).  A block owns 16 consecutive queries of one (request, head); its WV (8) waves take the 16-key tiles w, w + WV for
// BOTH products, so nothing but the row maxima, the row sums and the 16 x 64 partial outputs crosses a wave:
//   scores  = Q K^T   v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulate), 16 per key tile; every operand is requested up front
//                     straight in operand layout (lane = (row l % 16, k group l / 16) holds four float4 chunks 16 c + 4 (l / 16) of its
//                     row: the contraction index is permuted the same way on both sides, which a dot product does not see)
//   softmax           row maximum by a 16-lane butterfly + one LDS exchange across the waves; p = exp(s - max)
//   out     = P V     p goes through a wave-private LDS tile to become an A operand; V rows are float4 loads whose four components
//                     are the four 16-dim output tiles (dim = 4 (l % 16) + tile), so a lane's four accumulators are one float4 of the row
// and the partial outputs of the WV waves are summed in the fixed order w0 + w1 + ... and scaled by 1 / sum.  A (request, head, query)'s
// result depends on that request's rows only.
template <int DK, int WV>
__global__ __launch_bounds__(64 * WV) void k_flow_attn_mfma(FlowAttn a) {
    static_assert(DK == 64, "operand layout below is written for 64-dim heads");
    constexpr int KT = 16 / WV, PS = KT * 16 + 4, OS = DK + 4;      // WV waves x KT key tiles of 16: at most 256 keys
    __shared__ float red_mx[WV][16], red_sum[WV][16];
    __shared__ __attribute__((aligned(16))) float Ps[WV][16][PS];
    __shared__ __attribute__((aligned(16))) float Os[WV][16][OS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.x * 16, h = blockIdx.y, n = blockIdx.z;
    const int HD = a.H * DK, ld = 3 * HD, S = a.Tc + a.T, nt = (S + 15) >> 4;
    const float* cache = a.cache ? a.cache + (a.cidx ? (size_t)a.cidx[n] * a.cache_half_stride : a.B > 0 ? (size_t)(n / a.B) * a.cache_half_stride : 0) +
                                       (size_t)h * a.Tcap * 2 * DK : nullptr;
    auto krow = [&](int j) -> const float* {      // K row of key j (its V row: + DK in the cache, + HD in qkv)
        return j < a.Tc ? cache + (size_t)flow_cache_row(j, a.prefix, a.ring_head, a.Tcap) * 2 * DK
                        : a.qkv + ((size_t)n * a.T + (j - a.Tc)) * ld + HD + h * DK;
    };
    const int iq = i0 + fr < a.T ? i0 + fr : a.T - 1;
    const float* qrow = a.qkv + ((size_t)n * a.T + iq) * ld + h * DK + 4 * kq;
    float4 q[4], k[KT][4], v[KT][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = *reinterpret_cast<const float4*>(qrow + 16 * c);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int kt = t * WV + wave;
#pragma unroll
        for (int c = 0; c < 4; ++c) { k[t][c] = make_float4(0.f, 0.f, 0.f, 0.f); v[t][c] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (kt < nt) {
            const int j = kt * 16 + fr;
            if (j < S) {
                const float* kr = krow(j) + 4 * kq;
#pragma unroll
                for (int c = 0; c < 4; ++c) k[t][c] = *reinterpret_cast<const float4*>(kr + 16 * c);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {      // B operand of P V: key 16 kt + 4 s4 + kq, dims 4 fr .. 4 fr + 3
                const int jv = kt * 16 + 4 * s4 + kq;
                if (jv < S) v[t][s4] = *reinterpret_cast<const float4*>(krow(jv) + (jv < a.Tc ? DK : HD) + 4 * fr);
            }
        }
    }
    f32x4 sc[KT];
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int kt = t * WV + wave;
        sc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kt < nt) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].x, k[t][c].x, sc[t], 0, 0, 0);
                sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].y, k[t][c].y, sc[t], 0, 0, 0);
                sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].z, k[t][c].z, sc[t], 0, 0, 0);
                sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c].w, k[t][c].w, sc[t], 0, 0, 0);
            }
        }
        const int j = kt * 16 + fr;      // accumulator r: query i0 + 4 kq + r, key j
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * kq + r;
            float s = sc[t][r] * a.scale;
            if (kt >= nt || j >= S || (a.mask_block > 0 && j > i && j / a.mask_block != i / a.mask_block)) s = -INFINITY;
            sc[t][r] = s;
            mx[r] = fmaxf(mx[r], s);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], off, 64));
        if (fr == 0) red_mx[wave][4 * kq + r] = mx[r];
    }
    __syncthreads();
    float sum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = 4 * kq + r;
        float gm = red_mx[0][qi];
#pragma unroll
        for (int w = 1; w < WV; ++w) gm = fmaxf(gm, red_mx[w][qi]);
        float sm = 0.0f;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float p = sc[t][r] == -INFINITY ? 0.0f : expf(sc[t][r] - gm);
            Ps[wave][qi][16 * t + fr] = p;
            sm += p;
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) sm += __shfl_xor(sm, off, 64);
        sum[r] = sm;
        if (fr == 0) red_sum[wave][qi] = sm;
    }
    __builtin_amdgcn_wave_barrier();      // the wave's own p tile: written in accumulator layout, read back as A operands
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (t * WV + wave < nt) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float pa = Ps[wave][fr][16 * t + 4 * s4 + kq];
                o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, v[t][s4].x, o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, v[t][s4].y, o[1], 0, 0, 0);
                o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, v[t][s4].z, o[2], 0, 0, 0);
                o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, v[t][s4].w, o[3], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)      // o[d][r] = out[query 4 kq + r][dim 4 fr + d]
        *reinterpret_cast<float4*>(&Os[wave][4 * kq + r][4 * fr]) = make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
    __syncthreads();
    const int qi = tid >> 4, c4 = (tid & 15) * 4;
    if (tid < 256 && i0 + qi < a.T) {
        float l = red_sum[0][qi];
        float4 o4 = *reinterpret_cast<const float4*>(&Os[0][qi][c4]);
#pragma unroll
        for (int w = 1; w < WV; ++w) {
            l += red_sum[w][qi];
            const float4 pw = *reinterpret_cast<const float4*>(&Os[w][qi][c4]);
            o4.x += pw.x; o4.y += pw.y; o4.z += pw.z; o4.w += pw.w;
        }
        const float inv = 1.0f / l;
        *reinterpret_cast<float4*>(a.out + ((size_t)n * a.T + i0 + qi) * HD + h * DK + c4) = make_float4(o4.x * inv, o4.y * inv, o4.z * inv, o4.w * inv);
    }
}
// VOX_FLOW_ATTN_MFMA=0: the VALU attention kernels for every call (A/B timing)
static bool flow_attn_mfma_on() {
    static const bool on = [] { const char* e = getenv("VOX_FLOW_ATTN_MFMA"); return !(e && e[0] == '0'); }();
    return on;
}
// VOX_FLOW_ATTN_TILE=0: k_flow_attn for every call (A/B timing)
static bool flow_attn_tile_on() {
    static const bool on = [] { const char* e = getenv("VOX_FLOW_ATTN_TILE"); return !(e && e[0] == '0'); }();
    return on;
}
static void launch_flow_attn(hipStream_t st, const FlowAttn& a, int T, int H, int N) {
    const int S = a.Tc + T;
    const size_t lds = ((size_t)S * (2 * a.dk + 4) + 4 * ((S + 3) & ~3) + 4 * a.dk) * sizeof(float);
    if (!a.P && !a.bu && a.dk == 64 && S <= 256 && flow_attn_mfma_on()) {
        static const bool wv4 = [] { const char* e = getenv("VOX_FLOW_ATTN_WV"); return e && atoi(e) == 4; }();      // A/B: four waves x four key tiles
        if (wv4) hipLaunchKernelGGL((k_flow_attn_mfma<64, 4>), dim3((T + 15) / 16, H, N), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_flow_attn_mfma<64, 8>), dim3((T + 15) / 16, H, N), dim3(512), 0, st, a);
        return;
    }
    if (!a.P && a.dk == 64 && lds <= 150 * 1024 && flow_attn_tile_on()) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_flow_attn_tile<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr = true;
        }
        int qg = 4;      // queries per block: one per wave while the grid stays within ~2 blocks per CU, more per wave beyond
        while ((T + qg - 1) / qg * H * N > 512 && qg < T) qg *= 2;
        hipLaunchKernelGGL(k_flow_attn_tile<64>, dim3((T + qg - 1) / qg, H, N), dim3(256), lds, st, a, qg);
        return;
    }
    hipLaunchKernelGGL(k_flow_attn, dim3((T + 3) / 4, H, N), dim3(256), 0, st, a);
}
// new K | V rows of request n -> cache [half = n][H][Tcap][2 dk], keeping the first `prefix` and the last Tcap - prefix of the T rows
__global__ __launch_bounds__(256) void k_flow_cache_store(const float* qkv, float* cache, int T, int H, int dk, int Tcap, int prefix,
                                                           size_t cache_half_stride, int N) {
    const int keep = T < Tcap ? T : Tcap;
    const size_t total = (size_t)N * H * keep * 2 * dk;
    const int HD = H * dk;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e = (int)(i % (2 * dk));
        size_t r = i / (2 * dk);
        const int j = (int)(r % keep); r /= keep;
        const int h = (int)(r % H), n = (int)(r / H);
        const int src = (T <= Tcap || j < prefix) ? j : T - (Tcap - j);
        const float v = qkv[((size_t)n * T + src) * 3 * HD + (e < dk ? HD : 2 * HD) + h * dk + (e < dk ? e : e - dk)];
        cache[(size_t)n * cache_half_stride + ((size_t)h * Tcap + j) * 2 * dk + e] = v;
    }
}
// evolving caches: the T new K | V rows of request row n are appended to its cache block cidx[n] — logical position len + t; positions
// below `prefix` are stored in place, the others in the ring of Tcap - prefix rows that starts at `head` (a row past the capacity lands on
// the oldest ring row: the sliding-window truncation of cosyvoice2.py:1016-1046 without moving the rows that stay)
__global__ __launch_bounds__(256) void k_flow_cache_append(const float* qkv, float* cache, const int* cidx, size_t stride, int T, int H, int dk,
                                                            int Tcap, int prefix, int len, int head, int N) {
    const size_t total = (size_t)N * H * T * 2 * dk;
    const int HD = H * dk, R = Tcap - prefix;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e = (int)(i % (2 * dk));
        size_t r = i / (2 * dk);
        const int t = (int)(r % T); r /= T;
        const int h = (int)(r % H), n = (int)(r / H);
        const int p = len + t;
        const int row = p < prefix ? p : prefix + (head + p - prefix) % R;
        const float v = qkv[((size_t)n * T + t) * 3 * HD + (e < dk ? HD : 2 * HD) + h * dk + (e < dk ? e : e - dk)];
        cache[(size_t)cidx[n] * stride + ((size_t)h * Tcap + row) * 2 * dk + e] = v;
    }
}
// last two rows of every request -> conv state [n][2][C] (slots != NULL: state row slots[n] instead of n)
__global__ __launch_bounds__(256) void k_flow_tail2(const float* x, float* st, int N, int T, int C, const int* slots = nullptr) {
    const size_t total = (size_t)N * 2 * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C), p = (int)((i / C) % 2), n = (int)(i / (2 * C));
        const int t = T - 2 + p;
        const float v = t >= 0 ? x[((size_t)n * T + t) * C + c] : 0.0f;
        st[slots ? ((size_t)slots[n] * 2 + p) * C + c : i] = v;
    }
}
// estimator input [2B * T][4 mel] = x | mu | spk | cond, the second (unconditional) half with mu = spk = cond = 0   (cosyvoice_flow.py:2737-2745)
__global__ __launch_bounds__(256) void k_flow_pack(const float* x, const float* mu, const float* spk, const float* cond, float* y, int B, int T,
                                                    int M, int spk_stride = 0) {
    const size_t total = (size_t)2 * B * T * 4 * M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (4 * M));
        const size_t row = i / (4 * M);
        const int n = (int)(row / T), t = (int)(row % T), b = n % B;
        const bool un = n >= B;
        const int part = c / M, cc = c % M;
        float v;
        if (part == 0) v = x[((size_t)b * T + t) * M + cc];
        else if (un) v = 0.0f;
        else if (part == 1) v = mu[((size_t)b * T + t) * M + cc];
        else if (part == 2) v = spk[(size_t)b * spk_stride + cc];
        else v = cond ? cond[((size_t)b * T + t) * M + cc] : 0.0f;
        y[i] = v;
    }
}
// x <- x + dt * ((1 + r) d[b] - r d[B + b])   (classifier-free guidance + Euler step, cosyvoice_flow.py:2774-2778)
__global__ __launch_bounds__(256) void k_flow_euler(float* x, const float* d, int B, int T, int M, float dt, float rate) {
    const size_t total = (size_t)B * T * M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float g = (1.0f + rate) * d[i] - rate * d[total + i];
        x[i] = x[i] + dt * g;
    }
}
// start noise z [M][T] (given, or the seeded stream) -> x [B][T][M]
__global__ __launch_bounds__(256) void k_flow_noise(const float* z, uint64_t seed, uint32_t stream, float* x, int B, int T, int M) {
    const size_t total = (size_t)B * T * M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % M), t = (int)((i / M) % T);
        float v;
        if (z) v = z[(size_t)c * T + t];
        else {
            uint32_t w0, w1;
            philox4((uint32_t)(c * T + t), stream, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), &w0, &w1);
            const float u1 = ((float)(w0 >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);
            v = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
        }
        x[i] = v;
    }
}
// [B][T][M] -> [B][M][T]
__global__ __launch_bounds__(256) void k_flow_to_bct(const float* x, float* y, int B, int T, int M) {
    const size_t total = (size_t)B * T * M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i % T), c = (int)((i / T) % M), b = (int)(i / ((size_t)T * M));
        y[i] = x[((size_t)b * T + t) * M + c];
    }
}
// L2-normalise the speaker embedding (F.normalize, eps 1e-12): one block
__global__ __launch_bounds__(256) void k_flow_l2norm(const float* x, float* y, int C) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < C; i += 256) s += x[i] * x[i];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    for (int i = threadIdx.x; i < C; i += 256) y[i] = x[i] / nrm;
}
// cond [rows][M]: the first n_feat rows = prompt_feat, the rest 0
__global__ __launch_bounds__(256) void k_flow_cond(const float* feat, float* cond, int rows, int n_feat, int M) {
    const size_t total = (size_t)rows * M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) cond[i] = (int)(i / M) < n_feat ? feat[i] : 0.0f;
}

// y[row] = a[row] | b[row]
__global__ __launch_bounds__(256) void k_flow_concat2(const float* a, const float* b, float* y, size_t rows, int C) {
    const size_t total = rows * 2 * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (2 * C));
        const size_t r = i / (2 * C);
        y[i] = c < C ? a[r * C + c] : b[r * C + c - C];
    }
}
__global__ void k_flow_slots(int* slots, int N, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) slots[i] = i / B;
}

struct vox_flow {
    vox_ctx* ctx;
    vox_flow_config cfg;
    vox_flow_weights w;
    std::vector<vox_flow_conformer_w> enc;
    std::vector<vox_flow_resnet_w> resnets;
    std::vector<vox_flow_tblock_w> tblocks;
    int max_batch, max_T, max_prompt_T, n_res, n_att;
    std::vector<float> dt;                  // host: Euler step sizes
    float* tb = nullptr;                    // [n_steps][n_res][C]: mlp_r(Mish(time_mlp(emb_s)))
    float* buf[10] = {};
    size_t buf_floats = 0;
    float *spk = nullptr, *pe = nullptr, *pp = nullptr;
    int* slots = nullptr;                   // [2 * max_batch]: conv-state slot (= guidance half) of every estimator request row
    // ResNet blocks: block1's causal conv with the block's residual 1-tap conv stacked behind it as C more output columns (its weight in
    // the tap that reads the current row, zeros in the other taps): one launch less per ResNet block, 140 per chunk
    std::vector<vox_conv_w> conv1x;
    std::vector<void*> conv1x_mem;
    // static prompt caches
    float *enc_kv = nullptr, *up_kv = nullptr, *att_kv = nullptr, *cnn1 = nullptr, *cnn2 = nullptr;
    int enc_len = 0, up_len = 0, att_len = 0, cnn1_w = 0;
    bool have_prompt = false;
    // per-request evolving caches (use_detokenizer_cache=True, cosyvoice2.py:1010-1083): slot-major copies of the five caches, each
    // slot starting as a copy of the prompt's; the K/V caches keep their first `prefix` rows and run the rest as a ring
    struct SlotState { int enc_len = 0, enc_head = 0, up_len = 0, up_head = 0, att_len = 0, att_head = 0; bool live = false; };
    int n_slots = 0;
    std::vector<SlotState> slot;
    float *s_enc = nullptr, *s_up = nullptr, *s_att = nullptr, *s_cnn1 = nullptr, *s_cnn2 = nullptr;
    int *s_eidx = nullptr, *s_cidx = nullptr;      // device: slot of request b [max_batch]; cache block of estimator row n [2 max_batch]
};

static inline int flow_res_cin(const vox_flow_config& c, int r) { return r == 0 ? 4 * c.mel : (r == 1 + c.est_mid ? 2 * c.est_ch : c.est_ch); }

extern "C" {

void vox_flow_destroy(vox_flow* m) {
    if (!m) return;
    for (int i = 0; i < 10; ++i) (void)hipFree(m->buf[i]);
    (void)hipFree(m->tb); (void)hipFree(m->spk); (void)hipFree(m->pe); (void)hipFree(m->pp); (void)hipFree(m->slots);
    (void)hipFree(m->enc_kv); (void)hipFree(m->up_kv); (void)hipFree(m->att_kv); (void)hipFree(m->cnn1); (void)hipFree(m->cnn2);
    (void)hipFree(m->s_enc); (void)hipFree(m->s_up); (void)hipFree(m->s_att); (void)hipFree(m->s_cnn1); (void)hipFree(m->s_cnn2);
    (void)hipFree(m->s_eidx); (void)hipFree(m->s_cidx);
    for (void* p : m->conv1x_mem) (void)hipFree(p);
    delete m;
}

}  // extern "C"

// Rows up to which the flows' GEMMs take the few-row kernels (k_rows_gemm where the shape is eligible, the LDS-staged 16-row kernel
// otherwise) and normalise inside them.  Unbounded by default: the few-row kernels tile any row count, and ONE kernel per shape at
// every batch size is what keeps a request's mel independent of the batch it shares (k_rows_gemm sums its K quarters in a different
// order than the staged kernels).  VOX_FLOW_ROWS=<n> restores a row bound (A/B timing).
static int flow_skinny_rows() {
    static const int v = [] { const char* e = getenv("VOX_FLOW_ROWS"); return e ? atoi(e) : (1 << 30); }();
    return v;
}
#define FLOW_SKINNY_ROWS flow_skinny_rows()
static const int FLOW_OFF0[4] = {0, 0, 0, 0};
static const int FLOW_OFF_C3[3] = {2, 1, 0};            // causal k3
static const int FLOW_OFF_C5[5] = {4, 3, 2, 1, 0};      // causal k5 (Upsample1D after its left pad of 4)
static const int FLOW_OFF_LA[8] = {0, -1, -2, -3, -4, -5, -6, -7};   // look-ahead k (PreLookaheadLayer.conv1)

// one conformer layer (norm_mha -> rel-pos attention -> residual; norm_ff -> SiLU FFN -> residual) on x [n * T][D] in place
// LayerNorm(w, b, eps) followed by a one-tap GEMM of the normalised rows.  Calls that take the few-row GEMM normalise inside its operand
// staging (one launch less per attention / feed-forward half); larger batches normalise into `scratch` first.
static int ln_gemm(hipStream_t st, const vox_conv_w& w, const float* x, const float* lw, const float* lb, float eps, float* scratch, int n, int L,
                   float* out, int gelu) {
    if (n * L <= g_conv_skinny_rows && w.n_taps == 1)
        return conv_gemm(st, w, x, nullptr, nullptr, n, L, 0, FLOW_OFF0, out, nullptr, nullptr, gelu, nullptr, nullptr, 0, nullptr, lw, lb, eps);
    hipLaunchKernelGGL(k_flow_ln, dim3((n * L + 3) / 4), dim3(256), 0, st, x, lw, lb, scratch, w.cin, eps, 1.0f, 0, (const float*)nullptr, 1, n * L);
    return conv_gemm(st, w, scratch, nullptr, nullptr, n, L, 0, FLOW_OFF0, out, nullptr, nullptr, gelu);
}
// evolving mode (eidx != NULL): `cache` is layer l of the slot-major cache, request b reads and appends to block eidx[b] (stride
// slot_stride) whose rows past `prefix` form a ring starting at `head`
static int flow_conformer(vox_flow* m, hipStream_t st, const vox_flow_conformer_w& w, float* x, int n, int T, const float* cache, int Tc, int Tcap,
                          float* store_cache, int store_cap, int store_prefix, float** B, const int* eidx = nullptr, size_t slot_stride = 0,
                          int head = 0) {
    const vox_flow_config& c = m->cfg;
    const int D = c.dim, H = c.enc_heads, dk = D / H, S = Tc + T;
    if (S > FLOW_MAXKEYS) return vox_fail(VOX_ERR_INVALID, "flow: %d keys > %d", S, FLOW_MAXKEYS);
    float *nrm = B[0], *qkv = B[1], *att = B[2];
    VOX_TRY(ln_gemm(st, w.qkv, x, w.ln_mha_w, w.ln_mha_b, 1e-12f, nrm, n, T, qkv, 0));
    VOX_TRY(conv_gemm(st, w.pos, m->pe, nullptr, nullptr, 1, 2 * S - 1, 0, FLOW_OFF0, m->pp, nullptr, nullptr, 0));
    FlowAttn a{qkv, cache, m->pp, w.bias_u, w.bias_v, att, T, H, dk, Tc, Tcap, 0, eidx ? slot_stride : 0, 1.0f / sqrtf((float)dk)};
    a.cidx = eidx; a.prefix = store_prefix; a.ring_head = head;
    launch_flow_attn(st, a, T, H, n);
    if (eidx)
        hipLaunchKernelGGL(k_flow_cache_append, dim3(ew_grid((size_t)n * H * T * 2 * dk)), dim3(256), 0, st, qkv, const_cast<float*>(cache), eidx,
                           slot_stride, T, H, dk, Tcap, store_prefix, Tc, head, n);
    else if (store_cache)
        hipLaunchKernelGGL(k_flow_cache_store, dim3(ew_grid((size_t)H * store_cap * 2 * dk)), dim3(256), 0, st, qkv, store_cache, T, H, dk, store_cap,
                           store_prefix, (size_t)0, 1);
    VOX_TRY(conv_gemm(st, w.out, att, nullptr, nullptr, n, T, 0, FLOW_OFF0, x, x, nullptr, 0));
    VOX_TRY(ln_gemm(st, w.w1, x, w.ln_ff_w, w.ln_ff_b, 1e-12f, nrm, n, T, qkv, 0));
    hipLaunchKernelGGL(k_flow_act, dim3(ew_grid((size_t)n * T * c.enc_ffn)), dim3(256), 0, st, qkv, (size_t)n * T * c.enc_ffn, 2);
    VOX_TRY(conv_gemm(st, w.w2, qkv, nullptr, nullptr, n, T, 0, FLOW_OFF0, x, x, nullptr, 0));
    return VOX_OK;
}

// tokens [B][T] -> mel frames x [B][2T][mel] (time-major, in m->buf[9]); init: B == 1, caches are written instead of read
// evolve != NULL (decode only): the per-request caches of the given slot state (all requests of the call share one state: equal cache
// lengths, as the reference's batched caches require) are read through m->s_eidx / m->s_cidx and appended to
static int flow_run(vox_flow* m, hipStream_t st, const int32_t* tokens, int B, int T, bool init, const float* prompt_feat, int n_feat,
                    const float* noise, uint64_t seed, uint32_t nstream, float* mu_out, const vox_flow::SlotState* evolve = nullptr) {
    const vox_flow_config& c = m->cfg;
    const vox_flow_weights& w = m->w;
    const int D = c.dim, M = c.mel, C = c.est_ch, T2 = 2 * T, H = c.enc_heads, dk = D / H, inner = c.est_heads * c.est_head_dim;
    float** Bf = m->buf;
    g_conv_planes = 3;
    struct RowsGuard { ~RowsGuard() { g_conv_skinny_rows = 48; g_conv_rows_gemm = false; } } rows_guard;      // also on the error returns
    g_conv_skinny_rows = FLOW_SKINNY_ROWS; g_conv_rows_gemm = true;
    // ---- encoder ----
    float* x = Bf[3];
    hipLaunchKernelGGL(k_flow_embed, dim3(ew_grid((size_t)B * T * D)), dim3(256), 0, st, tokens, w.embedding, Bf[0], (size_t)B * T, D);
    VOX_TRY(conv_gemm(st, w.embed_lin, Bf[0], nullptr, nullptr, B, T, 0, FLOW_OFF0, Bf[1], nullptr, nullptr, 0));
    hipLaunchKernelGGL(k_flow_ln, dim3((B * T + 3) / 4), dim3(256), 0, st, Bf[1], w.embed_ln_w, w.embed_ln_b, x, D, 1e-5f, sqrtf((float)D), 0, nullptr, 1, B * T);
    // PreLookaheadLayer (empty context): conv k(pre+1) looking ahead, leaky_relu, causal conv k3, residual
    VOX_TRY(conv_gemm(st, w.pre1, x, nullptr, nullptr, B, T, 0, FLOW_OFF_LA, Bf[0], nullptr, nullptr, 0));
    hipLaunchKernelGGL(k_flow_act, dim3(ew_grid((size_t)B * T * D)), dim3(256), 0, st, Bf[0], (size_t)B * T * D, 3);
    VOX_TRY(conv_gemm(st, w.pre2, Bf[0], nullptr, nullptr, B, T, 0, FLOW_OFF_C3, x, x, nullptr, 0));
    {
        const int Tc = init ? 0 : (evolve ? evolve->enc_len : m->enc_len), S = Tc + T, cap = c.max_cache / 2;
        hipLaunchKernelGGL(k_flow_relpos, dim3(ew_grid((size_t)(2 * S - 1) * D)), dim3(256), 0, st, m->pe, S, D);
        const size_t lay = (size_t)H * cap * 2 * dk;
        for (int l = 0; l < c.enc_layers; ++l) {
            float* kv = (evolve ? m->s_enc : m->enc_kv) + (size_t)l * lay;
            VOX_TRY(flow_conformer(m, st, m->enc[l], x, B, T, init ? nullptr : kv, Tc, cap, init ? kv : nullptr, cap, c.prefix / 2, Bf,
                                   evolve ? m->s_eidx : nullptr, (size_t)c.enc_layers * lay, evolve ? evolve->enc_head : 0));
        }
        if (init) m->enc_len = T < cap ? T : cap;
    }
    // Upsample1D (nearest x2, causal conv k5), up_embed, the second conformer stack, after_norm, encoder_proj
    hipLaunchKernelGGL(k_flow_repeat2, dim3(ew_grid((size_t)B * T2 * D)), dim3(256), 0, st, x, Bf[0], (size_t)B * T, D);
    VOX_TRY(conv_gemm(st, w.up_conv, Bf[0], nullptr, nullptr, B, T2, 0, FLOW_OFF_C5, Bf[1], nullptr, nullptr, 0));
    VOX_TRY(conv_gemm(st, w.up_embed_lin, Bf[1], nullptr, nullptr, B, T2, 0, FLOW_OFF0, Bf[0], nullptr, nullptr, 0));
    hipLaunchKernelGGL(k_flow_ln, dim3((B * T2 + 3) / 4), dim3(256), 0, st, Bf[0], w.up_embed_ln_w, w.up_embed_ln_b, x, D, 1e-5f, sqrtf((float)D), 0, nullptr, 1, B * T2);
    {
        const int Tc = init ? 0 : (evolve ? evolve->up_len : m->up_len), S = Tc + T2, cap = c.max_cache;
        hipLaunchKernelGGL(k_flow_relpos, dim3(ew_grid((size_t)(2 * S - 1) * D)), dim3(256), 0, st, m->pe, S, D);
        const size_t lay = (size_t)H * cap * 2 * dk;
        for (int l = 0; l < c.up_layers; ++l) {
            float* kv = (evolve ? m->s_up : m->up_kv) + (size_t)l * lay;
            VOX_TRY(flow_conformer(m, st, m->enc[c.enc_layers + l], x, B, T2, init ? nullptr : kv, Tc, cap, init ? kv : nullptr, cap, c.prefix, Bf,
                                   evolve ? m->s_eidx : nullptr, (size_t)c.up_layers * lay, evolve ? evolve->up_head : 0));
        }
        if (init) m->up_len = T2 < cap ? T2 : cap;
    }
    hipLaunchKernelGGL(k_flow_ln, dim3((B * T2 + 3) / 4), dim3(256), 0, st, x, w.after_w, w.after_b, Bf[0], D, 1e-5f, 1.0f, 0, nullptr, 1, B * T2);
    float* mu = Bf[8];
    VOX_TRY(conv_gemm(st, w.enc_proj, Bf[0], nullptr, nullptr, B, T2, 0, FLOW_OFF0, mu, nullptr, nullptr, 0));
    if (mu_out) (void)hipMemcpyAsync(mu_out, mu, (size_t)B * T2 * M * 4, hipMemcpyDeviceToDevice, st);
    // ---- conditional flow matching: n_steps Euler steps of the estimator on the doubled (guidance) batch ----
    float *xs = Bf[9], *cond = Bf[7];
    if (init) hipLaunchKernelGGL(k_flow_cond, dim3(ew_grid((size_t)T2 * M)), dim3(256), 0, st, prompt_feat, cond, T2, n_feat, M);
    hipLaunchKernelGGL(k_flow_noise, dim3(ew_grid((size_t)B * T2 * M)), dim3(256), 0, st, noise, seed, nstream, xs, B, T2, M);
    const int N = 2 * B, capA = c.max_cache, hd = c.est_head_dim, HE = c.est_heads;
    const size_t att_layer = (size_t)HE * capA * 2 * hd, att_half = (size_t)c.n_steps * m->n_att * att_layer;
    const int Tc = init ? 0 : (evolve ? evolve->att_len : m->att_len);
    if (Tc + T2 > FLOW_MAXKEYS) return vox_fail(VOX_ERR_INVALID, "flow: chunk too long");
    hipLaunchKernelGGL(k_flow_slots, dim3((N + 63) / 64), dim3(64), 0, st, m->slots, N, B);
    // conv-state row / attention-cache block of estimator row n: the guidance half (shared prompt caches) or 2 slot + half (evolving)
    const int* crow = evolve ? m->s_cidx : m->slots;
    const int srows = evolve ? 2 * m->n_slots : 2;             // state rows per (step, resnet) block
    for (int s = 0; s < c.n_steps; ++s) {
        float *hA = Bf[0], *a1 = Bf[1], *a2 = Bf[2], *a3 = Bf[3], *skip = Bf[4], *cat = Bf[5], *hB = Bf[6];
        hipLaunchKernelGGL(k_flow_pack, dim3(ew_grid((size_t)N * T2 * 4 * M)), dim3(256), 0, st, xs, mu, m->spk, init ? cond : nullptr, cat, B, T2, M);
        const float* in = cat;
        float* h = hA;
        int li = 0;
        for (int r = 0; r < m->n_res; ++r) {
            const vox_flow_resnet_w& rw = m->resnets[r];
            const int cin = flow_res_cin(c, r);
            if (r == m->n_res - 1) {      // up block: pack([x, skip]) along channels
                hipLaunchKernelGGL(k_flow_concat2, dim3(ew_grid((size_t)N * T2 * 2 * C)), dim3(256), 0, st, in, skip, cat, (size_t)N * T2, C);
                in = cat;
            }
            h = (in == hA) ? hB : hA;
            float* st1 = (evolve ? m->s_cnn1 : m->cnn1) + ((size_t)s * m->n_res + r) * srows * 2 * m->cnn1_w;
            float* st2 = (evolve ? m->s_cnn2 : m->cnn2) + ((size_t)s * m->n_res + r) * srows * 2 * C;
            if (init) hipLaunchKernelGGL(k_flow_tail2, dim3(ew_grid((size_t)N * 2 * cin)), dim3(256), 0, st, in, st1, N, T2, cin, (const int*)nullptr);
            // block1: (cached) causal conv k3 -> LayerNorm -> Mish, + the time projection; block2 likewise; + res_conv(x)
            const bool fold = !m->conv1x.empty();      // a1 = [block1 conv | res_conv(in)], 2 C wide
            VOX_TRY(conv_gemm(st, fold ? m->conv1x[r] : rw.conv1, in, init ? nullptr : st1, crow, N, T2, 2, FLOW_OFF_C3, a1, nullptr, nullptr, 0));
            if (evolve) hipLaunchKernelGGL(k_flow_tail2, dim3(ew_grid((size_t)N * 2 * cin)), dim3(256), 0, st, in, st1, N, T2, cin, crow);
            hipLaunchKernelGGL(k_flow_ln, dim3((N * T2 + 3) / 4), dim3(256), 0, st, a1, rw.ln1_w, rw.ln1_b, a2, C, 1e-5f, 1.0f, 1,
                               m->tb + ((size_t)s * m->n_res + r) * C, N * T2, N * T2, fold ? 2 * C : 0, 0);
            if (init) hipLaunchKernelGGL(k_flow_tail2, dim3(ew_grid((size_t)N * 2 * C)), dim3(256), 0, st, a2, st2, N, T2, C, (const int*)nullptr);
            VOX_TRY(conv_gemm(st, rw.conv2, a2, init ? nullptr : st2, crow, N, T2, 2, FLOW_OFF_C3, fold ? a3 : a1, nullptr, nullptr, 0));
            if (evolve) hipLaunchKernelGGL(k_flow_tail2, dim3(ew_grid((size_t)N * 2 * C)), dim3(256), 0, st, a2, st2, N, T2, C, crow);
            if (fold) {      // h = Mish(LayerNorm(block2 conv)) + res_conv(in)
                hipLaunchKernelGGL(k_flow_ln, dim3((N * T2 + 3) / 4), dim3(256), 0, st, a3, rw.ln2_w, rw.ln2_b, h, C, 1e-5f, 1.0f, 1, a1 + C, 1, N * T2, 0,
                                   2 * C);
            } else {
                hipLaunchKernelGGL(k_flow_ln, dim3((N * T2 + 3) / 4), dim3(256), 0, st, a1, rw.ln2_w, rw.ln2_b, a2, C, 1e-5f, 1.0f, 1, nullptr, 1, N * T2);
                VOX_TRY(conv_gemm(st, rw.res, in, nullptr, nullptr, N, T2, 0, FLOW_OFF0, h, a2, nullptr, 0));       // h = block2 + res_conv(in)
            }
            for (int j = 0; j < c.est_blocks; ++j, ++li) {
                const vox_flow_tblock_w& tw = m->tblocks[li];
                float* kv = (evolve ? m->s_att : m->att_kv) + ((size_t)s * m->n_att + li) * att_layer;
                VOX_TRY(ln_gemm(st, tw.qkv, h, tw.ln1_w, tw.ln1_b, 1e-5f, a1, N, T2, a2, 0));
                FlowAttn a{a2, init ? nullptr : kv, nullptr, nullptr, nullptr, a3, T2, HE, hd, Tc, capA, B, att_half, 1.0f / sqrtf((float)hd)};
                if (evolve) { a.cidx = m->s_cidx; a.prefix = c.prefix; a.ring_head = evolve->att_head; }
                launch_flow_attn(st, a, T2, HE, N);
                if (evolve)
                    hipLaunchKernelGGL(k_flow_cache_append, dim3(ew_grid((size_t)N * HE * T2 * 2 * hd)), dim3(256), 0, st, a2, kv, m->s_cidx, att_half,
                                       T2, HE, hd, capA, c.prefix, Tc, evolve->att_head, N);
                if (init)
                    hipLaunchKernelGGL(k_flow_cache_store, dim3(ew_grid((size_t)N * HE * capA * 2 * hd)), dim3(256), 0, st, a2, kv, T2, HE, hd, capA,
                                       c.prefix, att_half, N);
                VOX_TRY(conv_gemm(st, tw.out, a3, nullptr, nullptr, N, T2, 0, FLOW_OFF0, h, h, nullptr, 0));
                VOX_TRY(ln_gemm(st, tw.ff1, h, tw.ln3_w, tw.ln3_b, 1e-5f, a1, N, T2, a2, 1));
                VOX_TRY(conv_gemm(st, tw.ff2, a2, nullptr, nullptr, N, T2, 0, FLOW_OFF0, h, h, nullptr, 0));
            }
            in = h;
            if (r == 0) {                  // down block: keep the skip, then its (uncached) causal conv
                (void)hipMemcpyAsync(skip, h, (size_t)N * T2 * C * 4, hipMemcpyDeviceToDevice, st);
                float* o = (h == hA) ? hB : hA;
                VOX_TRY(conv_gemm(st, w.down_conv, skip, nullptr, nullptr, N, T2, 0, FLOW_OFF_C3, o, nullptr, nullptr, 0));
                in = o;
            }
        }
        VOX_TRY(conv_gemm(st, w.up_conv2, in, nullptr, nullptr, N, T2, 0, FLOW_OFF_C3, a1, nullptr, nullptr, 0));
        VOX_TRY(conv_gemm(st, w.final_conv, a1, nullptr, nullptr, N, T2, 0, FLOW_OFF_C3, a2, nullptr, nullptr, 0));
        hipLaunchKernelGGL(k_flow_ln, dim3((N * T2 + 3) / 4), dim3(256), 0, st, a2, w.final_ln_w, w.final_ln_b, a1, C, 1e-5f, 1.0f, 1, nullptr, 1, N * T2);
        VOX_TRY(conv_gemm(st, w.final_proj, a1, nullptr, nullptr, N, T2, 0, FLOW_OFF0, a2, nullptr, nullptr, 0));
        hipLaunchKernelGGL(k_flow_euler, dim3(ew_grid((size_t)B * T2 * M)), dim3(256), 0, st, xs, a2, B, T2, M, m->dt[s], c.cfg_rate);
    }
    if (init) m->att_len = T2 < capA ? T2 : capA;
    g_conv_skinny_rows = 48; g_conv_rows_gemm = false;
    return VOX_OK;
}

extern "C" {

int vox_flow_create(vox_ctx* ctx, const vox_flow_config* cfg, const vox_flow_weights* w, int max_batch, int max_T, int max_prompt_T,
                    const float* time_emb, const float* dt, vox_flow** out) {
    if (!ctx || !cfg || !w || !out || !time_emb || !dt) return vox_fail(VOX_ERR_INVALID, "flow_create: NULL");
    const vox_flow_config& c = *cfg;
    if (c.dim % c.enc_heads || c.dim / c.enc_heads > 128 || c.est_head_dim > 128 || c.dim % 32 || c.est_ch % 32 || (4 * c.mel) % 32 ||
        (c.est_heads * c.est_head_dim) % 32 || c.enc_ffn % 32 || c.spk_dim % 32 || c.pre_lookahead + 1 > 8 || max_batch < 1 || max_T < 1 ||
        max_prompt_T < 1 || c.n_steps < 1)
        return vox_fail(VOX_ERR_INVALID, "flow_create: bad config");
    vox_flow* m = new vox_flow();
    m->ctx = ctx; m->cfg = c; m->w = *w; m->max_batch = max_batch; m->max_T = max_T; m->max_prompt_T = max_prompt_T;
    m->n_res = 2 + c.est_mid; m->n_att = m->n_res * c.est_blocks;
    m->enc.assign(w->enc, w->enc + c.enc_layers + c.up_layers);
    m->resnets.assign(w->resnets, w->resnets + m->n_res);
    m->tblocks.assign(w->tblocks, w->tblocks + m->n_att);
    m->dt.assign(dt, dt + c.n_steps);
    const int D = c.dim, C = c.est_ch, TE = 4 * C, inner = c.est_heads * c.est_head_dim, dk = D / c.enc_heads;
    size_t rows = (size_t)2 * max_batch * 2 * max_T;
    if ((size_t)2 * 2 * (max_prompt_T + 3) > rows) rows = (size_t)2 * 2 * (max_prompt_T + 3);
    size_t wmax = 3 * (size_t)D;
    for (size_t v : {(size_t)c.enc_ffn, (size_t)4 * C, (size_t)3 * inner, (size_t)4 * c.mel, (size_t)TE}) wmax = v > wmax ? v : wmax;
    m->buf_floats = rows * wmax;
    m->cnn1_w = 4 * c.mel > 2 * C ? 4 * c.mel : 2 * C;
    const size_t capE = c.max_cache / 2, capU = c.max_cache;
    bool ok = true;
    auto alloc = [&](float** p, size_t n) { ok = ok && hipMalloc((void**)p, n * 4) == hipSuccess; };
    for (int i = 0; i < 10; ++i) alloc(&m->buf[i], m->buf_floats);
    alloc(&m->tb, (size_t)c.n_steps * m->n_res * C);
    alloc(&m->spk, c.mel);
    alloc(&m->pe, (size_t)2 * FLOW_MAXKEYS * D);
    alloc(&m->pp, (size_t)2 * FLOW_MAXKEYS * D);
    alloc(&m->enc_kv, (size_t)c.enc_layers * c.enc_heads * capE * 2 * dk);
    alloc(&m->up_kv, (size_t)c.up_layers * c.enc_heads * capU * 2 * dk);
    alloc(&m->att_kv, (size_t)2 * c.n_steps * m->n_att * c.est_heads * c.max_cache * 2 * c.est_head_dim);
    alloc(&m->cnn1, (size_t)c.n_steps * m->n_res * 4 * m->cnn1_w);
    alloc(&m->cnn2, (size_t)c.n_steps * m->n_res * 4 * C);
    ok = ok && hipMalloc((void**)&m->slots, (size_t)2 * (max_batch > 1 ? max_batch : 1) * 4) == hipSuccess;
    static const bool resfold = [] { const char* e = getenv("VOX_FLOW_RESFOLD"); return !(e && e[0] == '0'); }();
    for (int r = 0; r < m->n_res && ok && resfold; ++r) {
        const vox_flow_resnet_w& rw = m->resnets[r];
        if (rw.conv1.n_taps != 3 || rw.res.n_taps != 1 || rw.conv1.n != C || rw.res.n != C || rw.conv1.cin != rw.res.cin) { m->conv1x.clear(); break; }
        const size_t cin = rw.conv1.cin, blk = (size_t)C * cin * 2;      // bytes of one tap's [C][cin] bf16 block
        char* wx = nullptr; float* bx = nullptr;
        ok = hipMalloc((void**)&wx, 3 * 2 * blk) == hipSuccess && hipMalloc((void**)&bx, (size_t)2 * C * 4) == hipSuccess;
        if (wx) m->conv1x_mem.push_back(wx);
        if (bx) m->conv1x_mem.push_back(bx);
        if (!ok) break;
        (void)hipMemset(wx, 0, 3 * 2 * blk); (void)hipMemset(bx, 0, (size_t)2 * C * 4);
        for (int t = 0; t < 3; ++t) (void)hipMemcpy(wx + (size_t)t * 2 * blk, (const char*)rw.conv1.w + (size_t)t * blk, blk, hipMemcpyDeviceToDevice);
        (void)hipMemcpy(wx + (size_t)2 * 2 * blk + blk, rw.res.w, blk, hipMemcpyDeviceToDevice);      // FLOW_OFF_C3: tap 2 reads the current row
        if (rw.conv1.bias) (void)hipMemcpy(bx, rw.conv1.bias, (size_t)C * 4, hipMemcpyDeviceToDevice);
        if (rw.res.bias) (void)hipMemcpy(bx + C, rw.res.bias, (size_t)C * 4, hipMemcpyDeviceToDevice);
        vox_conv_w cx = rw.conv1;
        cx.w = wx; cx.bias = bx; cx.n = 2 * C; cx.bias_mod = 0;
        m->conv1x.push_back(cx);
    }
    if (!ok) { vox_flow_destroy(m); return vox_fail(VOX_ERR_NOMEM, "flow_create: hipMalloc failed"); }
    // time projections of every Euler step and resnet, once: tb[s][r] = mlp_r(Mish(linear_2(SiLU(linear_1(emb_s)))))
    hipStream_t st = nullptr;
    g_conv_planes = 3;
    int rc = VOX_OK;
    (void)hipMemcpy(m->buf[0], time_emb, (size_t)c.n_steps * 4 * c.mel * 4, hipMemcpyHostToDevice);
    rc = conv_gemm(st, w->time1, m->buf[0], nullptr, nullptr, 1, c.n_steps, 0, FLOW_OFF0, m->buf[1], nullptr, nullptr, 0);
    hipLaunchKernelGGL(k_flow_act, dim3(ew_grid((size_t)c.n_steps * TE)), dim3(256), 0, st, m->buf[1], (size_t)c.n_steps * TE, 2);
    if (rc == VOX_OK) rc = conv_gemm(st, w->time2, m->buf[1], nullptr, nullptr, 1, c.n_steps, 0, FLOW_OFF0, m->buf[2], nullptr, nullptr, 0);
    hipLaunchKernelGGL(k_flow_act, dim3(ew_grid((size_t)c.n_steps * TE)), dim3(256), 0, st, m->buf[2], (size_t)c.n_steps * TE, 1);
    for (int r = 0; r < m->n_res && rc == VOX_OK; ++r) {
        rc = conv_gemm(st, m->resnets[r].mlp, m->buf[2], nullptr, nullptr, 1, c.n_steps, 0, FLOW_OFF0, m->buf[3], nullptr, nullptr, 0);
        (void)hipMemcpy2DAsync(m->tb + (size_t)r * C, (size_t)m->n_res * C * 4, m->buf[3], (size_t)C * 4, (size_t)C * 4, c.n_steps,
                               hipMemcpyDeviceToDevice, st);
    }
    if (hipStreamSynchronize(st) != hipSuccess || rc != VOX_OK) { vox_flow_destroy(m); return rc != VOX_OK ? rc : vox_fail(VOX_ERR_HIP, "flow_create: time MLP failed"); }
    *out = m;
    return VOX_OK;
}

int vox_flow_set_prompt(vox_flow* m, void* stream, const int32_t* prompt_tokens, int n_prompt, const float* prompt_feat,
                        const float* embedding, const float* noise, uint64_t seed, uint32_t noise_stream, float* prompt_mel) {
    if (!m || !prompt_tokens || !prompt_feat || !embedding) return vox_fail(VOX_ERR_INVALID, "flow_set_prompt: NULL");
    if (n_prompt < 3 || n_prompt > m->max_prompt_T) return vox_fail(VOX_ERR_INVALID, "flow_set_prompt: %d prompt tokens out of range", n_prompt);
    hipStream_t st = (hipStream_t)stream;
    const vox_flow_config& c = m->cfg;
    const int T = n_prompt + 3;
    // speaker vector: spk_embed_affine_layer(normalize(embedding))
    hipLaunchKernelGGL(k_flow_l2norm, dim3(1), dim3(256), 0, st, embedding, m->buf[0], c.spk_dim);
    g_conv_planes = 3;
    VOX_TRY(conv_gemm(st, m->w.spk, m->buf[0], nullptr, nullptr, 1, 1, 0, FLOW_OFF0, m->spk, nullptr, nullptr, 0));
    // tokens = prompt + its first three again (cosyvoice2.py:864-867)
    int* tok = reinterpret_cast<int*>(m->buf[6]);
    (void)hipMemcpyAsync(tok, prompt_tokens, (size_t)n_prompt * 4, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(tok + n_prompt, prompt_tokens, (size_t)3 * 4, hipMemcpyDeviceToDevice, st);
    m->have_prompt = false;
    VOX_TRY(flow_run(m, st, tok, 1, T, true, prompt_feat, 2 * n_prompt, noise, seed, noise_stream, nullptr));
    if (prompt_mel) hipLaunchKernelGGL(k_flow_to_bct, dim3(ew_grid((size_t)2 * T * c.mel)), dim3(256), 0, st, m->buf[9], prompt_mel, 1, 2 * T, c.mel);
    m->have_prompt = true;
    return VOX_OK;
}

int vox_flow_decode_chunk(vox_flow* m, void* stream, const int32_t* tokens, int n, int T, const float* noise, uint64_t seed,
                          uint32_t noise_stream, float* mel, float* mu) {
    if (!m || !tokens || !mel) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk: NULL");
    if (!m->have_prompt) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk: no prompt set (vox_flow_set_prompt)");
    if (n < 1 || n > m->max_batch || T < 1 || T > m->max_T) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk: n %d / T %d out of range", n, T);
    hipStream_t st = (hipStream_t)stream;
    VOX_TRY(flow_run(m, st, tokens, n, T, false, nullptr, 0, noise, seed, noise_stream, mu));
    hipLaunchKernelGGL(k_flow_to_bct, dim3(ew_grid((size_t)n * 2 * T * m->cfg.mel)), dim3(256), 0, st, m->buf[9], mel, n, 2 * T, m->cfg.mel);
    return VOX_OK;
}

// ---- per-request evolving caches (CosyVoice2Decoder with shared_prompt_cache_mode=False; cosyvoice2.py:1010-1083) --------------------
int vox_flow_enable_slots(vox_flow* m, int n_slots) {
    if (!m || n_slots < 1) return vox_fail(VOX_ERR_INVALID, "flow_enable_slots: bad arguments");
    if (m->n_slots) return m->n_slots == n_slots ? VOX_OK : vox_fail(VOX_ERR_INVALID, "flow_enable_slots: already enabled with %d slots", m->n_slots);
    const vox_flow_config& c = m->cfg;
    const int D = c.dim, C = c.est_ch, dk = D / c.enc_heads;
    const size_t capE = c.max_cache / 2, capU = c.max_cache;
    const size_t e = (size_t)c.enc_layers * c.enc_heads * capE * 2 * dk, u = (size_t)c.up_layers * c.enc_heads * capU * 2 * dk;
    const size_t a = (size_t)2 * c.n_steps * m->n_att * c.est_heads * c.max_cache * 2 * c.est_head_dim;
    bool ok = true;
    auto alloc = [&](float** p, size_t n) { ok = ok && hipMalloc((void**)p, n * 4) == hipSuccess; };
    alloc(&m->s_enc, e * n_slots); alloc(&m->s_up, u * n_slots); alloc(&m->s_att, a * n_slots);
    alloc(&m->s_cnn1, (size_t)c.n_steps * m->n_res * 2 * n_slots * 2 * m->cnn1_w);
    alloc(&m->s_cnn2, (size_t)c.n_steps * m->n_res * 2 * n_slots * 2 * C);
    ok = ok && hipMalloc((void**)&m->s_eidx, (size_t)m->max_batch * 4) == hipSuccess && hipMalloc((void**)&m->s_cidx, (size_t)2 * m->max_batch * 4) == hipSuccess;
    if (!ok) return vox_fail(VOX_ERR_NOMEM, "flow_enable_slots: hipMalloc failed (%zu MB per slot)", (e + u + a) * 4 >> 20);
    m->n_slots = n_slots;
    m->slot.assign(n_slots, vox_flow::SlotState{});
    return VOX_OK;
}

// a request takes slot `slot`: its caches start as a copy of the prompt's (model/cosyvoice2.py:514-560: the per-request cache is the
// expanded initial cache of the default speaker)
int vox_flow_slot_reset(vox_flow* m, void* stream, int slot) {
    if (!m || slot < 0 || slot >= m->n_slots) return vox_fail(VOX_ERR_INVALID, "flow_slot_reset: slot %d out of range", slot);
    if (!m->have_prompt) return vox_fail(VOX_ERR_INVALID, "flow_slot_reset: no prompt set (vox_flow_set_prompt)");
    hipStream_t st = (hipStream_t)stream;
    const vox_flow_config& c = m->cfg;
    const int D = c.dim, C = c.est_ch, dk = D / c.enc_heads;
    const size_t e = (size_t)c.enc_layers * c.enc_heads * (c.max_cache / 2) * 2 * dk, u = (size_t)c.up_layers * c.enc_heads * c.max_cache * 2 * dk;
    const size_t a = (size_t)2 * c.n_steps * m->n_att * c.est_heads * c.max_cache * 2 * c.est_head_dim;
    VOX_HIP(hipMemcpyAsync(m->s_enc + e * slot, m->enc_kv, e * 4, hipMemcpyDeviceToDevice, st));
    VOX_HIP(hipMemcpyAsync(m->s_up + u * slot, m->up_kv, u * 4, hipMemcpyDeviceToDevice, st));
    VOX_HIP(hipMemcpyAsync(m->s_att + a * slot, m->att_kv, a * 4, hipMemcpyDeviceToDevice, st));
    // conv states: per (step, resnet) block [2 halves][2][cin] -> rows 2 slot, 2 slot + 1 of the slot-major block
    const size_t rows = (size_t)c.n_steps * m->n_res;
    for (int r = 0; r < m->n_res; ++r) {
        const int cin = flow_res_cin(c, r);
        VOX_HIP(hipMemcpy2DAsync(m->s_cnn1 + (size_t)r * 2 * m->n_slots * 2 * m->cnn1_w + (size_t)2 * slot * 2 * cin, (size_t)m->n_res * 2 * m->n_slots * 2 * m->cnn1_w * 4,
                                 m->cnn1 + (size_t)r * 4 * m->cnn1_w, (size_t)m->n_res * 4 * m->cnn1_w * 4, (size_t)4 * cin * 4, c.n_steps,
                                 hipMemcpyDeviceToDevice, st));
        VOX_HIP(hipMemcpy2DAsync(m->s_cnn2 + (size_t)r * 2 * m->n_slots * 2 * C + (size_t)2 * slot * 2 * C, (size_t)m->n_res * 2 * m->n_slots * 2 * C * 4,
                                 m->cnn2 + (size_t)r * 4 * C, (size_t)m->n_res * 4 * C * 4, (size_t)4 * C * 4, c.n_steps, hipMemcpyDeviceToDevice, st));
    }
    (void)rows;
    vox_flow::SlotState& ss = m->slot[slot];
    ss = vox_flow::SlotState{};
    ss.enc_len = m->enc_len; ss.up_len = m->up_len; ss.att_len = m->att_len; ss.live = true;
    return VOX_OK;
}

int vox_flow_slot_state(vox_flow* m, int slot, int32_t out[6]) {
    if (!m || !out || slot < 0 || slot >= m->n_slots) return vox_fail(VOX_ERR_INVALID, "flow_slot_state: bad arguments");
    const vox_flow::SlotState& x = m->slot[slot];
    out[0] = x.enc_len; out[1] = x.up_len; out[2] = x.att_len; out[3] = x.enc_head; out[4] = x.up_head; out[5] = x.att_head;
    return VOX_OK;
}

struct FlowSlotIdx {
    static constexpr int MAXN = 64;
    int e[MAXN], c[2 * MAXN];
};
__global__ void k_flow_slot_idx(FlowSlotIdx ix, int n, int* eidx, int* cidx) {
    const int t = threadIdx.x;
    if (t < n) eidx[t] = ix.e[t];
    else if (t >= FlowSlotIdx::MAXN && t - FlowSlotIdx::MAXN < 2 * n) cidx[t - FlowSlotIdx::MAXN] = ix.c[t - FlowSlotIdx::MAXN];
}

// one chunk of n requests that own slots[0..n): the flow runs against their caches, which then take this chunk's rows (sliding
// window: the first `prefix` rows + the most recent ones).  The requests of one call must share their cache lengths, as the reference's
// batched cache tensors do (DecoderCache.cat); the caller groups them.
int vox_flow_decode_chunk_slots(vox_flow* m, void* stream, const int32_t* tokens, int n, int T, const int32_t* slots, const float* noise,
                                uint64_t seed, uint32_t noise_stream, float* mel, float* mu) {
    if (!m || !tokens || !mel || !slots) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk_slots: NULL");
    if (!m->n_slots) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk_slots: vox_flow_enable_slots was not called");
    if (n < 1 || n > m->max_batch || T < 1 || T > m->max_T) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk_slots: n %d / T %d out of range", n, T);
    const vox_flow_config& c = m->cfg;
    std::vector<int> eidx(n), cidx(2 * n);
    for (int b = 0; b < n; ++b) {
        const int sl = slots[b];
        if (sl < 0 || sl >= m->n_slots || !m->slot[sl].live) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk_slots: slot %d not initialised", sl);
        for (int b2 = 0; b2 < b; ++b2)
            if (slots[b2] == sl) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk_slots: slot %d twice in one call", sl);
        const vox_flow::SlotState &x = m->slot[sl], &y = m->slot[slots[0]];
        if (x.enc_len != y.enc_len || x.enc_head != y.enc_head || x.up_len != y.up_len || x.up_head != y.up_head || x.att_len != y.att_len ||
            x.att_head != y.att_head)
            return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk_slots: the requests of a call must have equal cache states (vox_flow_slot_state)");
        eidx[b] = sl; cidx[b] = 2 * sl; cidx[n + b] = 2 * sl + 1;
    }
    const int capE = c.max_cache / 2, capU = c.max_cache;
    if (T > capE - c.prefix / 2 || 2 * T > capU - c.prefix) return vox_fail(VOX_ERR_INVALID, "flow_decode_chunk_slots: chunk longer than the cache ring");
    hipStream_t st = (hipStream_t)stream;
    if (n <= FlowSlotIdx::MAXN) {
        // the slot index vectors travel as kernel arguments (copied at launch): no host buffer whose lifetime the stream depends on,
        // so the call returns without waiting for the chunks queued before it.  Host-side slot states advance at enqueue time: all
        // calls on one flow object must go through ONE stream (they are ordered against each other only there).
        FlowSlotIdx ix{};
        for (int b = 0; b < n; ++b) { ix.e[b] = eidx[b]; ix.c[b] = cidx[b]; ix.c[n + b] = cidx[n + b]; }
        hipLaunchKernelGGL(k_flow_slot_idx, dim3(1), dim3(3 * FlowSlotIdx::MAXN), 0, st, ix, n, m->s_eidx, m->s_cidx);
    } else {
        VOX_HIP(hipMemcpyAsync(m->s_eidx, eidx.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        VOX_HIP(hipMemcpyAsync(m->s_cidx, cidx.data(), (size_t)2 * n * 4, hipMemcpyHostToDevice, st));
        VOX_HIP(hipStreamSynchronize(st));          // (the index vectors above are stack-owned host memory)
    }
    const vox_flow::SlotState cur = m->slot[slots[0]];
    VOX_TRY(flow_run(m, st, tokens, n, T, false, nullptr, 0, noise, seed, noise_stream, mu, &cur));
    hipLaunchKernelGGL(k_flow_to_bct, dim3(ew_grid((size_t)n * 2 * T * c.mel)), dim3(256), 0, st, m->buf[9], mel, n, 2 * T, c.mel);
    auto advance = [](int& len, int& head, int add, int cap, int prefix) {
        const int over = len + add - cap;
        if (over > 0) head = (head + over) % (cap - prefix);
        len = len + add < cap ? len + add : cap;
    };
    for (int b = 0; b < n; ++b) {
        vox_flow::SlotState& x = m->slot[slots[b]];
        advance(x.enc_len, x.enc_head, T, capE, c.prefix / 2);
        advance(x.up_len, x.up_head, 2 * T, capU, c.prefix);
        advance(x.att_len, x.att_head, 2 * T, capU, c.prefix);
    }
    return VOX_OK;
}

}  // extern "C"

// fade_in_out (tokenizer/cosyvoice2.py:46-54): the first `fade` samples of every row cross-fade with the tail of the previous chunk
// (prev NULL = silence), computed in double like the reference's float64 Hamming window
__global__ __launch_bounds__(256) void k_fade_in(float* wav, const float* prev, const double* win, int n, int L, int fade) {
    const size_t total = (size_t)n * fade;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / fade), t = (int)(i % fade);
        const double v = (double)wav[(size_t)b * L + t] * win[t] + (prev ? (double)prev[(size_t)b * fade + t] : 0.0) * win[fade + t];
        wav[(size_t)b * L + t] = (float)v;
    }
}
extern "C" int vox_fade_in_out(void* stream, float* wav, int n, int L, const float* prev_tail, const double* window, int fade) {
    if (!wav || !window || n < 1 || fade < 1 || fade > L) return vox_fail(VOX_ERR_INVALID, "fade_in_out: bad arguments");
    hipLaunchKernelGGL(k_fade_in, dim3(ew_grid((size_t)n * fade)), dim3(256), 0, (hipStream_t)stream, wav, prev_tail, window, n, L, fade);
    return VOX_OK;
}

// the seeded CFM start noise as a tensor [mel][frames] (what vox_flow_* draw when `noise` is NULL): lets a caller that replays a captured
// graph keep the draw outside it (a graph freezes scalar arguments such as the stream id)
__global__ __launch_bounds__(256) void k_flow_noise_z(uint64_t seed, uint32_t stream, float* z, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        uint32_t w0, w1;
        philox4((uint32_t)i, stream, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), &w0, &w1);
        const float u1 = ((float)(w0 >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);
        z[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
}
extern "C" int vox_flow_fill_noise(void* stream, uint64_t seed, uint32_t noise_stream, int mel, int frames, float* z) {
    if (!z || mel < 1 || frames < 1) return vox_fail(VOX_ERR_INVALID, "flow_fill_noise: bad arguments");
    hipLaunchKernelGGL(k_flow_noise_z, dim3(ew_grid((size_t)mel * frames)), dim3(256), 0, (hipStream_t)stream, seed, noise_stream, z, (size_t)mel * frames);
    return VOX_OK;
}

// ====================================================================================================================
// GLM-4-Voice flow (speech tokens -> mel): block conformer encoder, length regulator, non-causal U-Net CFM.  include/voxhip.h has the contract.
// ====================================================================================================================
// GroupNorm over (channels of a group) x (all rows of a request) + Mish + optional per-channel addend; x [N][T][ld], C real channels
// (columns >= C are written 0); one block per (group, request)
__global__ __launch_bounds__(256) void k_flow_groupnorm(const float* x, const float* w, const float* b, float* y, int T, int C, int ld, int G,
                                                         float eps, const float* add) {
    __shared__ float red[8];
    const int g = blockIdx.x, n = blockIdx.y, cg = C / G;
    const float* xb = x + (size_t)n * T * ld;
    float* yb = y + (size_t)n * T * ld;
    const int cnt = T * cg;
    if (256 % cg == 0 && T <= 48 * (256 / cg)) {
        // the group fits the block's registers (GLM: 32 channels x <= 384 frames): element i = tid + 256 k is (frame tid / cg + k * 256 / cg,
        // channel tid % cg) — the general path's element -> thread assignment and accumulation order without its two integer divisions
        // per element and per pass, and with ONE read of x instead of three (the kernel is 16-128 blocks of pure latency)
        constexpr int KM = 48;
        const int c = g * cg + threadIdx.x % cg, r0 = threadIdx.x / cg, rs = 256 / cg;
        float xv[KM];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int r = r0 + k * rs;
            xv[k] = r < T ? xb[(size_t)r * ld + c] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (r0 + k * rs < T) s += xv[k];
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)cnt;
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (r0 + k * rs < T) { const float d = xv[k] - mean; v += d * d; }
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = v;
        __syncthreads();
        const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)cnt + eps);
        const float wc = w[c], bc = b[c], ac = add ? add[c] : 0.0f;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int r = r0 + k * rs;
            if (r < T) {
                float o = mish_f((xv[k] - mean) * rstd * wc + bc);
                if (add) o += ac;
                yb[(size_t)r * ld + c] = o;
            }
        }
        if (g == G - 1 && ld > C)
            for (int i = threadIdx.x; i < T * (ld - C); i += 256) yb[(size_t)(i / (ld - C)) * ld + C + i % (ld - C)] = 0.0f;
        return;
    }
    float s = 0.0f;
    for (int i = threadIdx.x; i < cnt; i += 256) s += xb[(size_t)(i / cg) * ld + g * cg + i % cg];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)cnt;
    float v = 0.0f;
    for (int i = threadIdx.x; i < cnt; i += 256) { const float d = xb[(size_t)(i / cg) * ld + g * cg + i % cg] - mean; v += d * d; }
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = v;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)cnt + eps);
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int c = g * cg + i % cg;
        const size_t o = (size_t)(i / cg) * ld + c;
        float r = mish_f((xb[o] - mean) * rstd * w[c] + b[c]);
        if (add) r += add[c];
        yb[o] = r;
    }
    if (g == G - 1 && ld > C)
        for (int i = threadIdx.x; i < T * (ld - C); i += 256) yb[(size_t)(i / (ld - C)) * ld + C + i % (ld - C)] = 0.0f;
}
// F.interpolate(mode="nearest") along time: y[b][t] = x[b][min(floor(t * T / To), T - 1)]
__global__ __launch_bounds__(256) void k_flow_interp(const float* x, float* y, int B, int T, int To, int ld) {
    const size_t total = (size_t)B * To * ld;
    const float sc = (float)T / (float)To;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % ld), t = (int)((i / ld) % To), b = (int)(i / ((size_t)ld * To));
        int src = (int)floorf((float)t * sc);
        src = src > T - 1 ? T - 1 : src;
        y[i] = x[((size_t)b * T + src) * ld + c];
    }
}
// Conv1d(k 3, stride 2, padding 1) as a 1-tap GEMM: y[n][t] = x[n][2t - 1] | x[n][2t] | x[n][2t + 1]  (zeros outside)
__global__ __launch_bounds__(256) void k_flow_stride2(const float* x, float* y, int N, int T, int To, int C) {
    const size_t total = (size_t)N * To * 3 * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C), j = (int)((i / C) % 3), t = (int)((i / (3 * (size_t)C)) % To), n = (int)(i / (3 * (size_t)C * To));
        const int ts = 2 * t - 1 + j;
        y[i] = (ts >= 0 && ts < T) ? x[((size_t)n * T + ts) * C + c] : 0.0f;
    }
}
// per-request start noise z [B][M][T] (given, or stream first + b) -> x [B][T][M]
__global__ __launch_bounds__(256) void k_flow_noise_req(const float* z, uint64_t seed, uint32_t first, float* x, int B, int T, int M) {
    const size_t total = (size_t)B * T * M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % M), t = (int)((i / M) % T), b = (int)(i / ((size_t)M * T));
        float v;
        if (z) v = z[((size_t)b * M + c) * T + t];
        else {
            uint32_t w0, w1;
            philox4((uint32_t)(c * T + t), first + (uint32_t)b, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), &w0, &w1);
            const float u1 = ((float)(w0 >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);
            v = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
        }
        x[i] = v;
    }
}
// rows of x [n][C] L2-normalised (F.normalize, eps 1e-12): one block per row; x NULL = zeros
__global__ __launch_bounds__(256) void k_flow_l2norm_rows(const float* x, float* y, int C) {
    __shared__ float red[4];
    const float* xr = x ? x + (size_t)blockIdx.x * C : nullptr;
    float s = 0.0f;
    for (int i = threadIdx.x; i < C; i += 256) { const float v = xr ? xr[i] : 0.0f; s += v * v; }
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    for (int i = threadIdx.x; i < C; i += 256) y[(size_t)blockIdx.x * C + i] = (xr ? xr[i] : 0.0f) / nrm;
}

struct vox_glmflow {
    vox_ctx* ctx;
    vox_glmflow_config cfg;
    vox_glmflow_weights w;
    std::vector<vox_flow_conformer_w> enc;
    std::vector<vox_flow_resnet_w> resnets;
    std::vector<vox_flow_tblock_w> tblocks;
    int max_batch, max_T, max_mel, n_res, n_att;
    std::vector<float> dt;
    float* tb = nullptr;
    float* buf[10] = {};
    size_t buf_floats = 0;
    float *spk = nullptr, *pe = nullptr, *pp = nullptr;
};
static const int FLOW_OFF_S3[3] = {1, 0, -1};           // "same" k3
static const int FLOW_OFF_T3[3] = {-1, 0, 1};           // ConvTranspose1d(4, 2, 1): taps d = -1, 0, 1 (tap d reads row t - d)

extern "C" {

void vox_glmflow_destroy(vox_glmflow* m) {
    if (!m) return;
    for (int i = 0; i < 10; ++i) (void)hipFree(m->buf[i]);
    (void)hipFree(m->tb); (void)hipFree(m->spk); (void)hipFree(m->pe); (void)hipFree(m->pp);
    delete m;
}

int vox_glmflow_create(vox_ctx* ctx, const vox_glmflow_config* cfg, const vox_glmflow_weights* w, int max_batch, int max_T, int max_mel,
                       const float* time_emb, const float* dt, vox_glmflow** out) {
    if (!ctx || !cfg || !w || !out || !time_emb || !dt) return vox_fail(VOX_ERR_INVALID, "glmflow_create: NULL");
    const vox_glmflow_config& c = *cfg;
    if (c.dim % c.enc_heads || c.dim / c.enc_heads > 128 || c.est_head_dim > 128 || c.dim % 32 || c.est_ch % 32 || (4 * c.mel) % 32 ||
        c.mel_padded % 32 || c.mel_padded < c.mel || (c.est_heads * c.est_head_dim) % 32 || c.enc_ffn % 32 || c.spk_dim % 32 ||
        c.est_ch % c.groups || max_batch < 1 || max_T < 1 || max_mel < 2 || c.n_steps < 1 || c.reg_layers > 4 || max_mel > FLOW_MAXKEYS ||
        max_T > FLOW_MAXKEYS)
        return vox_fail(VOX_ERR_INVALID, "glmflow_create: bad config");
    vox_glmflow* m = new vox_glmflow();
    m->ctx = ctx; m->cfg = c; m->w = *w; m->max_batch = max_batch; m->max_T = max_T; m->max_mel = max_mel;
    m->n_res = 4 + c.est_mid; m->n_att = m->n_res * c.est_blocks;
    m->enc.assign(w->enc, w->enc + c.enc_layers);
    m->resnets.assign(w->resnets, w->resnets + m->n_res);
    m->tblocks.assign(w->tblocks, w->tblocks + m->n_att);
    m->dt.assign(dt, dt + c.n_steps);
    const int D = c.dim, C = c.est_ch, TE = 4 * C, inner = c.est_heads * c.est_head_dim;
    size_t rows = (size_t)2 * max_batch * max_mel;
    if ((size_t)max_batch * max_T > rows) rows = (size_t)max_batch * max_T;
    size_t wmax = 3 * (size_t)D;
    for (size_t v : {(size_t)c.enc_ffn, (size_t)4 * C, (size_t)3 * inner, (size_t)4 * c.mel, (size_t)TE, (size_t)3 * C}) wmax = v > wmax ? v : wmax;
    m->buf_floats = rows * wmax;
    bool ok = true;
    auto alloc = [&](float** p, size_t n) { ok = ok && hipMalloc((void**)p, n * 4) == hipSuccess; };
    for (int i = 0; i < 10; ++i) alloc(&m->buf[i], m->buf_floats);
    alloc(&m->tb, (size_t)c.n_steps * m->n_res * C);
    alloc(&m->spk, (size_t)max_batch * c.mel);
    alloc(&m->pe, (size_t)2 * FLOW_MAXKEYS * D);
    alloc(&m->pp, (size_t)2 * FLOW_MAXKEYS * D);
    if (!ok) { vox_glmflow_destroy(m); return vox_fail(VOX_ERR_NOMEM, "glmflow_create: hipMalloc failed"); }
    hipStream_t st = nullptr;
    g_conv_planes = 3;
    g_conv_skinny_rows = 48; g_conv_rows_gemm = false;
    int rc = VOX_OK;
    (void)hipMemcpy(m->buf[0], time_emb, (size_t)c.n_steps * 4 * c.mel * 4, hipMemcpyHostToDevice);
    rc = conv_gemm(st, w->time1, m->buf[0], nullptr, nullptr, 1, c.n_steps, 0, FLOW_OFF0, m->buf[1], nullptr, nullptr, 0);
    hipLaunchKernelGGL(k_flow_act, dim3(ew_grid((size_t)c.n_steps * TE)), dim3(256), 0, st, m->buf[1], (size_t)c.n_steps * TE, 2);
    if (rc == VOX_OK) rc = conv_gemm(st, w->time2, m->buf[1], nullptr, nullptr, 1, c.n_steps, 0, FLOW_OFF0, m->buf[2], nullptr, nullptr, 0);
    hipLaunchKernelGGL(k_flow_act, dim3(ew_grid((size_t)c.n_steps * TE)), dim3(256), 0, st, m->buf[2], (size_t)c.n_steps * TE, 1);
    for (int r = 0; r < m->n_res && rc == VOX_OK; ++r) {
        rc = conv_gemm(st, m->resnets[r].mlp, m->buf[2], nullptr, nullptr, 1, c.n_steps, 0, FLOW_OFF0, m->buf[3], nullptr, nullptr, 0);
        (void)hipMemcpy2DAsync(m->tb + (size_t)r * C, (size_t)m->n_res * C * 4, m->buf[3], (size_t)C * 4, (size_t)C * 4, c.n_steps,
                               hipMemcpyDeviceToDevice, st);
    }
    if (hipStreamSynchronize(st) != hipSuccess || rc != VOX_OK) { vox_glmflow_destroy(m); return rc != VOX_OK ? rc : vox_fail(VOX_ERR_HIP, "glmflow_create: time MLP failed"); }
    *out = m;
    return VOX_OK;
}

int vox_glmflow_decode(vox_glmflow* m, void* stream, const int32_t* tokens, int n, int T, int Tm, const float* embedding, const float* noise,
                       uint64_t seed, uint32_t first_stream, float* mel) {
    if (!m || !tokens || !mel) return vox_fail(VOX_ERR_INVALID, "glmflow_decode: NULL");
    if (n < 1 || n > m->max_batch || T < 1 || T > m->max_T || Tm < 2 || Tm > m->max_mel)
        return vox_fail(VOX_ERR_INVALID, "glmflow_decode: n %d / T %d / mel frames %d out of range", n, T, Tm);
    hipStream_t st = (hipStream_t)stream;
    const vox_glmflow_config& c = m->cfg;
    const vox_glmflow_weights& w = m->w;
    const int D = c.dim, M = c.mel, MP = c.mel_padded, C = c.est_ch, H = c.enc_heads, dk = D / H, HE = c.est_heads, hd = c.est_head_dim;
    float** Bf = m->buf;
    g_conv_planes = 3;
    struct RowsGuard { ~RowsGuard() { g_conv_skinny_rows = 48; g_conv_rows_gemm = false; } } rows_guard;      // also on the error returns
    g_conv_skinny_rows = FLOW_SKINNY_ROWS; g_conv_rows_gemm = true;
    // speaker vector per request: spk_embed_affine_layer(normalize(embedding)); GLM-4-Voice passes zeros (glm.py:2647)
    hipLaunchKernelGGL(k_flow_l2norm_rows, dim3(n), dim3(256), 0, st, embedding, Bf[0], c.spk_dim);
    VOX_TRY(conv_gemm(st, w.spk, Bf[0], nullptr, nullptr, n, 1, 0, FLOW_OFF0, m->spk, nullptr, nullptr, 0));
    // ---- encoder: embed, LinearNoSubsampling, block-masked relative-position conformer layers, after_norm, encoder_proj ----
    float* x = Bf[3];
    hipLaunchKernelGGL(k_flow_embed, dim3(ew_grid((size_t)n * T * D)), dim3(256), 0, st, tokens, w.embedding, Bf[0], (size_t)n * T, D);
    VOX_TRY(conv_gemm(st, w.embed_lin, Bf[0], nullptr, nullptr, n, T, 0, FLOW_OFF0, Bf[1], nullptr, nullptr, 0));
    hipLaunchKernelGGL(k_flow_ln, dim3((n * T + 3) / 4), dim3(256), 0, st, Bf[1], w.embed_ln_w, w.embed_ln_b, x, D, 1e-5f, sqrtf((float)D), 0, nullptr, 1, n * T);
    hipLaunchKernelGGL(k_flow_relpos, dim3(ew_grid((size_t)(2 * T - 1) * D)), dim3(256), 0, st, m->pe, T, D);
    for (int l = 0; l < c.enc_layers; ++l) {
        const vox_flow_conformer_w& cw = m->enc[l];
        float *nrm = Bf[0], *qkv = Bf[1], *att = Bf[2];
        hipLaunchKernelGGL(k_flow_ln, dim3((n * T + 3) / 4), dim3(256), 0, st, x, cw.ln_mha_w, cw.ln_mha_b, nrm, D, 1e-12f, 1.0f, 0, nullptr, 1, n * T);
        VOX_TRY(conv_gemm(st, cw.qkv, nrm, nullptr, nullptr, n, T, 0, FLOW_OFF0, qkv, nullptr, nullptr, 0));
        VOX_TRY(conv_gemm(st, cw.pos, m->pe, nullptr, nullptr, 1, 2 * T - 1, 0, FLOW_OFF0, m->pp, nullptr, nullptr, 0));
        FlowAttn a{qkv, nullptr, m->pp, cw.bias_u, cw.bias_v, att, T, H, dk, 0, 0, 0, 0, 1.0f / sqrtf((float)dk), c.block_size};
        launch_flow_attn(st, a, T, H, n);
        VOX_TRY(conv_gemm(st, cw.out, att, nullptr, nullptr, n, T, 0, FLOW_OFF0, x, x, nullptr, 0));
        hipLaunchKernelGGL(k_flow_ln, dim3((n * T + 3) / 4), dim3(256), 0, st, x, cw.ln_ff_w, cw.ln_ff_b, nrm, D, 1e-12f, 1.0f, 0, nullptr, 1, n * T);
        VOX_TRY(conv_gemm(st, cw.w1, nrm, nullptr, nullptr, n, T, 0, FLOW_OFF0, qkv, nullptr, nullptr, 0));
        hipLaunchKernelGGL(k_flow_act, dim3(ew_grid((size_t)n * T * c.enc_ffn)), dim3(256), 0, st, qkv, (size_t)n * T * c.enc_ffn, 2);
        VOX_TRY(conv_gemm(st, cw.w2, qkv, nullptr, nullptr, n, T, 0, FLOW_OFF0, x, x, nullptr, 0));
    }
    hipLaunchKernelGGL(k_flow_ln, dim3((n * T + 3) / 4), dim3(256), 0, st, x, w.after_w, w.after_b, Bf[0], D, 1e-5f, 1.0f, 0, nullptr, 1, n * T);
    VOX_TRY(conv_gemm(st, w.enc_proj, Bf[0], nullptr, nullptr, n, T, 0, FLOW_OFF0, Bf[1], nullptr, nullptr, 0));       // [n*T][MP] (pad columns 0)
    // ---- length regulator: nearest resampling to Tm frames, (conv k3, GroupNorm(1), Mish) x reg_layers, conv k1 ----
    hipLaunchKernelGGL(k_flow_interp, dim3(ew_grid((size_t)n * Tm * MP)), dim3(256), 0, st, Bf[1], Bf[0], n, T, Tm, MP);
    for (int i = 0; i < c.reg_layers; ++i) {
        VOX_TRY(conv_gemm(st, w.reg_conv[i], Bf[0], nullptr, nullptr, n, Tm, 0, FLOW_OFF_S3, Bf[1], nullptr, nullptr, 0));
        hipLaunchKernelGGL(k_flow_groupnorm, dim3(1, n), dim3(256), 0, st, Bf[1], w.reg_gn_w[i], w.reg_gn_b[i], Bf[0], Tm, M, MP, 1, 1e-5f, nullptr);
    }
    float* mu = Bf[8];
    VOX_TRY(conv_gemm(st, w.reg_out, Bf[0], nullptr, nullptr, n, Tm, 0, FLOW_OFF0, mu, nullptr, nullptr, 0));           // [n*Tm][M]
    // ---- CFM: n_steps Euler steps of the U-Net on the doubled (classifier-free guidance) batch ----
    float* xs = Bf[9];
    hipLaunchKernelGGL(k_flow_noise_req, dim3(ew_grid((size_t)n * Tm * M)), dim3(256), 0, st, noise, seed, first_stream, xs, n, Tm, M);
    const int N = 2 * n, Th = (Tm - 1) / 2 + 1;
    auto tblocks = [&](float* h, int rows_T, int& li, float* a1, float* a2, float* a3) -> int {
        for (int j = 0; j < c.est_blocks; ++j, ++li) {
            const vox_flow_tblock_w& tw = m->tblocks[li];
            VOX_TRY(ln_gemm(st, tw.qkv, h, tw.ln1_w, tw.ln1_b, 1e-5f, a1, N, rows_T, a2, 0));
            FlowAttn a{a2, nullptr, nullptr, nullptr, nullptr, a3, rows_T, HE, hd, 0, 0, 0, 0, 1.0f / sqrtf((float)hd), 0};
            launch_flow_attn(st, a, rows_T, HE, N);
            VOX_TRY(conv_gemm(st, tw.out, a3, nullptr, nullptr, N, rows_T, 0, FLOW_OFF0, h, h, nullptr, 0));
            VOX_TRY(ln_gemm(st, tw.ff1, h, tw.ln3_w, tw.ln3_b, 1e-5f, a1, N, rows_T, a2, 1));
            VOX_TRY(conv_gemm(st, tw.ff2, a2, nullptr, nullptr, N, rows_T, 0, FLOW_OFF0, h, h, nullptr, 0));
        }
        return VOX_OK;
    };
    for (int s = 0; s < c.n_steps; ++s) {
        float *hA = Bf[0], *a1 = Bf[1], *a2 = Bf[2], *a3 = Bf[3], *skip0 = Bf[4], *cat = Bf[5], *hB = Bf[6], *skip1 = Bf[7];
        hipLaunchKernelGGL(k_flow_pack, dim3(ew_grid((size_t)N * Tm * 4 * M)), dim3(256), 0, st, xs, mu, m->spk, nullptr, cat, n, Tm, M, M);
        int li = 0, r = 0;
        // ResnetBlock1D (conv k3, GroupNorm(groups), Mish; + time projection; again; + res_conv) then the transformer blocks, in -> out
        auto group = [&](const float* in, float* o, int rows_T) -> int {
            const vox_flow_resnet_w& rw = m->resnets[r];
            VOX_TRY(conv_gemm(st, rw.conv1, in, nullptr, nullptr, N, rows_T, 0, FLOW_OFF_S3, a1, nullptr, nullptr, 0));
            hipLaunchKernelGGL(k_flow_groupnorm, dim3(c.groups, N), dim3(256), 0, st, a1, rw.ln1_w, rw.ln1_b, a2, rows_T, C, C, c.groups, 1e-5f,
                               m->tb + ((size_t)s * m->n_res + r) * C);
            VOX_TRY(conv_gemm(st, rw.conv2, a2, nullptr, nullptr, N, rows_T, 0, FLOW_OFF_S3, a1, nullptr, nullptr, 0));
            hipLaunchKernelGGL(k_flow_groupnorm, dim3(c.groups, N), dim3(256), 0, st, a1, rw.ln2_w, rw.ln2_b, a2, rows_T, C, C, c.groups, 1e-5f, nullptr);
            VOX_TRY(conv_gemm(st, rw.res, in, nullptr, nullptr, N, rows_T, 0, FLOW_OFF0, o, a2, nullptr, 0));
            ++r;
            return tblocks(o, rows_T, li, a1, a2, a3);
        };
        VOX_TRY(group(cat, skip0, Tm));                                                       // down 0 (its output is the outer skip)
        hipLaunchKernelGGL(k_flow_stride2, dim3(ew_grid((size_t)N * Th * 3 * C)), dim3(256), 0, st, skip0, cat, N, Tm, Th, C);
        VOX_TRY(conv_gemm(st, w.down_s2, cat, nullptr, nullptr, N, Th, 0, FLOW_OFF0, hA, nullptr, nullptr, 0));
        VOX_TRY(group(hA, skip1, Th));                                                        // down 1
        VOX_TRY(conv_gemm(st, w.down_conv1, skip1, nullptr, nullptr, N, Th, 0, FLOW_OFF_S3, hA, nullptr, nullptr, 0));
        float *cur = hA, *nxt = hB;
        for (int i = 0; i < c.est_mid; ++i) {
            VOX_TRY(group(cur, nxt, Th));
            float* t = cur; cur = nxt; nxt = t;
        }
        hipLaunchKernelGGL(k_flow_concat2, dim3(ew_grid((size_t)N * Th * 2 * C)), dim3(256), 0, st, cur, skip1, cat, (size_t)N * Th, C);
        VOX_TRY(group(cat, nxt, Th));                                                         // up 0
        VOX_TRY(conv_gemm(st, w.up_tconv, nxt, nullptr, nullptr, N, Th, 0, FLOW_OFF_T3, cur, nullptr, nullptr, 0));      // [N][2 Th][C]
        // x[:, :, :skip.shape[-1]]: the transposed conv gives 2 Th rows per request, the skip has Tm (<= 2 Th)
        if (2 * Th != Tm) {
            for (int q = 0; q < N; ++q)
                (void)hipMemcpyAsync(nxt + (size_t)q * Tm * C, cur + (size_t)q * 2 * Th * C, (size_t)Tm * C * 4, hipMemcpyDeviceToDevice, st);
            float* t = cur; cur = nxt; nxt = t;
        }
        hipLaunchKernelGGL(k_flow_concat2, dim3(ew_grid((size_t)N * Tm * 2 * C)), dim3(256), 0, st, cur, skip0, cat, (size_t)N * Tm, C);
        VOX_TRY(group(cat, nxt, Tm));                                                         // up 1
        VOX_TRY(conv_gemm(st, w.up_conv1, nxt, nullptr, nullptr, N, Tm, 0, FLOW_OFF_S3, a1, nullptr, nullptr, 0));
        VOX_TRY(conv_gemm(st, w.final_conv, a1, nullptr, nullptr, N, Tm, 0, FLOW_OFF_S3, a2, nullptr, nullptr, 0));
        hipLaunchKernelGGL(k_flow_groupnorm, dim3(c.groups, N), dim3(256), 0, st, a2, w.final_gn_w, w.final_gn_b, a1, Tm, C, C, c.groups, 1e-5f, nullptr);
        VOX_TRY(conv_gemm(st, w.final_proj, a1, nullptr, nullptr, N, Tm, 0, FLOW_OFF0, a2, nullptr, nullptr, 0));
        hipLaunchKernelGGL(k_flow_euler, dim3(ew_grid((size_t)n * Tm * M)), dim3(256), 0, st, xs, a2, n, Tm, M, m->dt[s], c.cfg_rate);
    }
    g_conv_skinny_rows = 48; g_conv_rows_gemm = false;
    hipLaunchKernelGGL(k_flow_to_bct, dim3(ew_grid((size_t)n * Tm * M)), dim3(256), 0, st, xs, mel, n, Tm, M);
    return VOX_OK;
}

}  // extern "C"

// ================================================================================================
// Prompt-side encoders of a voice-clone request (run once per request, before prefill)
// ================================================================================================
// ---- speaker encoder: log-mel front end + ECAPA-TDNN  (model/qwen3_tts.py:21-88, 317-532, 835-891) ----
// Layout: fp32 time-major [t][C].  A reflect-padded "same" conv is a centred-tap conv_gemm over a reflect-padded copy of its input
// ([T + 2p][C]); rows p .. p+T-1 of the result are the conv's output (contiguous, so later stages just offset the pointer).
#define ENC_MAX_FFT 2048
__global__ __launch_bounds__(256) void k_enc_logmel(const float* audio, int N, const float* window, const float* basis, int n_fft, int hop,
                                                     int n_mels, int MP, float* mel) {
    __shared__ double xs[ENC_MAX_FFT], tw[ENC_MAX_FFT];
    __shared__ float mag[ENC_MAX_FFT / 2 + 1];
    const int t = blockIdx.x, pad = (n_fft - hop) / 2, mask = n_fft - 1, nb = n_fft / 2 + 1;
    for (int j = threadIdx.x; j < n_fft; j += 256) {
        int i = t * hop + j - pad;                       // F.pad(..., mode="reflect"): no edge repeat
        i = i < 0 ? -i : (i >= N ? 2 * (N - 1) - i : i);
        xs[j] = (double)(audio[i] * window[j]);
        tw[j] = cospi(2.0 * j / n_fft);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += 256) {
        double re = 0.0, im = 0.0;
        const int q = 3 * n_fft / 4;                     // sin(a) = cos(a - pi/2)
        for (int n = 0; n < n_fft; ++n) {
            const int idx = (k * n) & mask;
            re += xs[n] * tw[idx];
            im += xs[n] * tw[(idx + q) & mask];
        }
        mag[k] = sqrtf((float)(re * re + im * im) + 1e-9f);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < MP; m += 256) {
        float v = 0.f;
        if (m < n_mels) {
            const float* b = basis + (size_t)m * nb;
            float acc = 0.f;
            for (int k = 0; k < nb; ++k) acc += b[k] * mag[k];
            v = logf(fmaxf(acc, 1e-5f));
        }
        mel[(size_t)t * MP + m] = v;
    }
}
// dst [T + 2p][C]: dst[r][c] = s1[refl(r - p)][c] (+ s2[refl(r - p)][c]), reflect without edge repeat
__global__ __launch_bounds__(256) void k_enc_gather(float* dst, int T, int p, int C, const float* s1, int ld1, const float* s2, int ld2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)(T + 2 * p) * C) return;
    const int r = (int)(i / C), c = (int)(i % C);
    int t = r - p;
    t = t < 0 ? -t : (t >= T ? 2 * (T - 1) - t : t);
    float v = s1[(size_t)t * ld1 + c];
    if (s2) v += s2[(size_t)t * ld2 + c];
    dst[i] = v;
}
__global__ __launch_bounds__(256) void k_enc_copy_cols(float* dst, int ldd, const float* src, int lds, int T, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    dst[(size_t)t * ldd + c] = src[(size_t)t * lds + c];
}
// mode 2: ReLU; 3: tanh(ReLU); 4: sigmoid
__global__ __launch_bounds__(256) void k_enc_act(float* x, size_t n, int mode) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    if (mode == 2) v = fmaxf(v, 0.f);
    else if (mode == 3) v = tanhf(fmaxf(v, 0.f));
    else if (mode == 4) v = 1.0f / (1.0f + expf(-v));
    x[i] = v;
}
// Per-channel weighted statistics over time (AttentiveStatisticsPooling._compute_statistics): weights 1/T (att == NULL) or the softmax over
// time of att[:, c].  mean[c] = sum w x, std[c] = sqrt(max(sum w (x - mean)^2, 1e-12)).  Block = 64 channels x 4 time lanes.
__global__ __launch_bounds__(256) void k_enc_colstats(const float* x, int T, int C, const float* att, float* mean, float* stdv) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    auto reduce = [&](float v, bool is_max) {
        red[tl][cl] = v;
        __syncthreads();
        float r = red[0][cl];
        for (int k = 1; k < 4; ++k) r = is_max ? fmaxf(r, red[k][cl]) : r + red[k][cl];
        __syncthreads();
        return r;
    };
    float amax = 0.f, inv = 1.0f / (float)T;
    if (att) {
        float mx = -INFINITY;
        if (ok) for (int t = tl; t < T; t += 4) mx = fmaxf(mx, att[(size_t)t * C + c]);
        amax = reduce(mx, true);
        float se = 0.f;
        if (ok) for (int t = tl; t < T; t += 4) se += expf(att[(size_t)t * C + c] - amax);
        inv = 1.0f / reduce(se, false);
    }
    float sm = 0.f;
    if (ok) for (int t = tl; t < T; t += 4) {
        const float w = att ? expf(att[(size_t)t * C + c] - amax) * inv : inv;
        sm += w * x[(size_t)t * C + c];
    }
    const float mu = reduce(sm, false);
    float sv = 0.f;
    if (ok) for (int t = tl; t < T; t += 4) {
        const float w = att ? expf(att[(size_t)t * C + c] - amax) * inv : inv;
        const float d = x[(size_t)t * C + c] - mu;
        sv += w * d * d;
    }
    const float var = reduce(sv, false);
    if (ok && tl == 0) {
        mean[c] = mu;
        if (stdv) stdv[c] = sqrtf(fmaxf(var, 1e-12f));
    }
}
// SE scale + residual (qwen3_tts.py:378, 532): out[t][c] = h[t][c] * s[c] + res[t][c]; a second copy goes into the aggregation buffer
__global__ __launch_bounds__(256) void k_enc_se_apply(const float* h, const float* s, const float* res, float* out, float* out2, int ld2, int T, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    const float v = h[i] * s[c] + res[i];
    out[i] = v;
    out2[(size_t)t * ld2 + c] = v;
}
// [x, mean, std] along channels (qwen3_tts.py:452-454): out [T][3C]
__global__ __launch_bounds__(256) void k_enc_asp_cat(const float* x, const float* mean, const float* stdv, float* out, int T, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * 3 * C) return;
    const int t = (int)(i / (3 * C)), c = (int)(i % (3 * C));
    out[i] = c < C ? x[(size_t)t * C + c] : (c < 2 * C ? mean[c - C] : stdv[c - 2 * C]);
}

struct vox_spkenc {
    vox_ctx* ctx;
    vox_spkenc_config cfg;
    vox_spkenc_weights w;
    int max_T, max_pad;
    float *mel = nullptr, *g = nullptr, *c0 = nullptr, *a = nullptr, *cat = nullptr, *h2 = nullptr, *hA = nullptr, *hB = nullptr,
          *co[8] = {}, *mfa_in = nullptr, *x = nullptr, *att_in = nullptr, *att_h = nullptr, *logits = nullptr, *vec = nullptr;
};
static inline unsigned enc_grid(size_t n) { return (unsigned)((n + 255) / 256); }
static inline void enc_act(hipStream_t st, float* x, size_t n, int mode) { hipLaunchKernelGGL(k_enc_act, dim3(enc_grid(n)), dim3(256), 0, st, x, n, mode); }

extern "C" {

void vox_spkenc_destroy(vox_spkenc* m) {
    if (!m) return;
    float* all[] = {m->mel, m->g, m->c0, m->a, m->cat, m->h2, m->hA, m->hB, m->mfa_in, m->x, m->att_in, m->att_h, m->logits, m->vec};
    for (float* p : all) if (p) (void)hipFree(p);
    for (float* p : m->co) if (p) (void)hipFree(p);
    delete m;
}

int vox_spkenc_create(vox_ctx* ctx, const vox_spkenc_config* cfg, const vox_spkenc_weights* w, int max_samples, vox_spkenc** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "spkenc_create: NULL");
    const vox_spkenc_config& c = *cfg;
    if (c.n_fft < 64 || c.n_fft > ENC_MAX_FFT || (c.n_fft & (c.n_fft - 1)) || c.hop < 1 || c.hop > c.n_fft || c.n_mels > c.n_mels_padded ||
        c.n_mels_padded % 32 || c.n_blocks < 1 || c.n_blocks > 4 || c.scale < 2 || c.scale > 8 || c.channels % (32 * c.scale) ||
        c.mfa_channels != c.n_blocks * c.channels || c.se_channels % 32 || c.att_channels % 32 || !(c.kernel0 & 1) || max_samples < c.n_fft)
        return vox_fail(VOX_ERR_INVALID, "spkenc_create: bad config");
    vox_spkenc* m = new vox_spkenc();
    m->ctx = ctx; m->cfg = c; m->w = *w;
    m->max_T = (max_samples + 2 * ((c.n_fft - c.hop) / 2) - c.n_fft) / c.hop + 1;
    int pad = c.dilation0 * (c.kernel0 - 1) / 2;
    for (int b = 0; b < c.n_blocks; ++b) {
        if (!(c.kernels[b] & 1)) { delete m; return vox_fail(VOX_ERR_INVALID, "spkenc_create: even kernel"); }
        pad = max(pad, c.dilations[b] * (c.kernels[b] - 1) / 2);
    }
    m->max_pad = pad;
    const size_t Tp = (size_t)m->max_T + 2 * pad, T = m->max_T, C = c.channels, M = c.mfa_channels, cw = C / c.scale;
    auto A = [&](float** p, size_t n) { return hipMalloc((void**)p, n * 4) == hipSuccess; };
    bool ok = A(&m->mel, T * c.n_mels_padded) && A(&m->g, Tp * (size_t)max(c.n_mels_padded, (int)cw)) && A(&m->c0, Tp * C) && A(&m->a, T * C) &&
              A(&m->cat, T * C) && A(&m->h2, T * C) && A(&m->hA, T * C) && A(&m->hB, T * C) && A(&m->mfa_in, T * M) && A(&m->x, T * M) &&
              A(&m->att_in, T * 3 * M) && A(&m->att_h, T * c.att_channels) && A(&m->logits, T * M) &&
              A(&m->vec, (size_t)(4 * M + 2 * C + c.se_channels + 64));
    for (int i = 1; i < c.scale && ok; ++i) ok = A(&m->co[i], Tp * cw);
    if (!ok) { vox_spkenc_destroy(m); return vox_fail(VOX_ERR_NOMEM, "spkenc_create: hipMalloc failed"); }
    *out = m;
    return VOX_OK;
}

int vox_spkenc_embed(vox_spkenc* m, void* stream, const float* audio, int n_samples, float* mel_out, int32_t* n_frames, float* emb_out) {
    if (!m || !audio || !emb_out) return vox_fail(VOX_ERR_INVALID, "spkenc_embed: NULL");
    const vox_spkenc_config& c = m->cfg;
    const vox_spkenc_weights& w = m->w;
    const int padw = (c.n_fft - c.hop) / 2;
    if (n_samples <= padw) return vox_fail(VOX_ERR_INVALID, "spkenc_embed: clip of %d samples is shorter than the STFT padding", n_samples);
    const int T = (n_samples + 2 * padw - c.n_fft) / c.hop + 1;
    if (T < 1 || T > m->max_T) return vox_fail(VOX_ERR_INVALID, "spkenc_embed: %d frames (capacity %d)", T, m->max_T);
    if (T <= m->max_pad) return vox_fail(VOX_ERR_INVALID, "spkenc_embed: %d frames, reflect padding needs more than %d", T, m->max_pad);
    if (n_frames) *n_frames = T;
    hipStream_t st = (hipStream_t)stream;
    const int saved_planes = g_conv_planes, saved_skinny = g_conv_skinny_rows;
    g_conv_planes = 3; g_conv_skinny_rows = 48; g_conv_rows_gemm = false;
    const int MP = c.n_mels_padded, C = c.channels, M = c.mfa_channels, cw = C / c.scale;
    int off[CG_MAXTAPS];
    static const int off0[1] = {0};
    hipLaunchKernelGGL(k_enc_logmel, dim3(T), dim3(256), 0, st, audio, n_samples, w.window, w.mel_basis, c.n_fft, c.hop, c.n_mels, MP, m->mel);
    if (mel_out) hipLaunchKernelGGL(k_enc_copy_cols, dim3(enc_grid((size_t)T * c.n_mels)), dim3(256), 0, st, mel_out, c.n_mels, m->mel, MP, T, c.n_mels);
    // blocks[0]: TDNN(mel -> C, k0, reflect)
    int p = c.dilation0 * (c.kernel0 - 1) / 2;
    hipLaunchKernelGGL(k_enc_gather, dim3(enc_grid((size_t)(T + 2 * p) * MP)), dim3(256), 0, st, m->g, T, p, MP, m->mel, MP, (const float*)nullptr, 0);
    hift_conv_offsets(c.kernel0, c.dilation0, off, 1);
    VOX_TRY(conv_gemm(st, w.conv0, m->g, nullptr, nullptr, 1, T + 2 * p, 0, off, m->c0, nullptr, nullptr, 0));
    enc_act(st, m->c0, (size_t)(T + 2 * p) * C, 2);
    const float* h = m->c0 + (size_t)p * C;
    float* hn = m->hA;
    float *vmean = m->vec, *vs1 = m->vec + C, *vs2 = vs1 + c.se_channels;       // SE vectors
    for (int b = 0; b < c.n_blocks; ++b) {
        const vox_spkenc_block_w& bw = w.blocks[b];
        VOX_TRY(conv_gemm(st, bw.tdnn1, h, nullptr, nullptr, 1, T, 0, off0, m->a, nullptr, nullptr, 0));
        enc_act(st, m->a, (size_t)T * C, 2);
        p = c.dilations[b] * (c.kernels[b] - 1) / 2;
        hift_conv_offsets(c.kernels[b], c.dilations[b], off, 1);
        for (int i = 1; i < c.scale; ++i) {                                       // Res2Net: chunk i sees chunk i + the previous output
            hipLaunchKernelGGL(k_enc_gather, dim3(enc_grid((size_t)(T + 2 * p) * cw)), dim3(256), 0, st, m->g, T, p, cw, m->a + i * cw, C,
                               i > 1 ? m->co[i - 1] + (size_t)p * cw : (const float*)nullptr, cw);
            VOX_TRY(conv_gemm(st, bw.res2[i - 1], m->g, nullptr, nullptr, 1, T + 2 * p, 0, off, m->co[i], nullptr, nullptr, 0));
            enc_act(st, m->co[i], (size_t)(T + 2 * p) * cw, 2);
        }
        hipLaunchKernelGGL(k_enc_copy_cols, dim3(enc_grid((size_t)T * cw)), dim3(256), 0, st, m->cat, C, m->a, C, T, cw);
        for (int i = 1; i < c.scale; ++i)
            hipLaunchKernelGGL(k_enc_copy_cols, dim3(enc_grid((size_t)T * cw)), dim3(256), 0, st, m->cat + i * cw, C, m->co[i] + (size_t)p * cw, cw, T, cw);
        VOX_TRY(conv_gemm(st, bw.tdnn2, m->cat, nullptr, nullptr, 1, T, 0, off0, m->h2, nullptr, nullptr, 0));
        enc_act(st, m->h2, (size_t)T * C, 2);
        hipLaunchKernelGGL(k_enc_colstats, dim3((C + 63) / 64), dim3(256), 0, st, m->h2, T, C, (const float*)nullptr, vmean, (float*)nullptr);
        VOX_TRY(conv_gemm(st, bw.se1, vmean, nullptr, nullptr, 1, 1, 0, off0, vs1, nullptr, nullptr, 0));
        enc_act(st, vs1, c.se_channels, 2);
        VOX_TRY(conv_gemm(st, bw.se2, vs1, nullptr, nullptr, 1, 1, 0, off0, vs2, nullptr, nullptr, 0));
        enc_act(st, vs2, C, 4);
        hipLaunchKernelGGL(k_enc_se_apply, dim3(enc_grid((size_t)T * C)), dim3(256), 0, st, m->h2, vs2, h, hn, m->mfa_in + (size_t)b * C, M, T, C);
        h = hn;
        hn = hn == m->hA ? m->hB : m->hA;
    }
    VOX_TRY(conv_gemm(st, w.mfa, m->mfa_in, nullptr, nullptr, 1, T, 0, off0, m->x, nullptr, nullptr, 0));
    enc_act(st, m->x, (size_t)T * M, 2);
    float *amean = m->vec + 2 * C + c.se_channels, *astd = amean + M, *pooled = astd + M;   // pooled = [mean | std], 2M
    hipLaunchKernelGGL(k_enc_colstats, dim3((M + 63) / 64), dim3(256), 0, st, m->x, T, M, (const float*)nullptr, amean, astd);
    hipLaunchKernelGGL(k_enc_asp_cat, dim3(enc_grid((size_t)T * 3 * M)), dim3(256), 0, st, m->x, amean, astd, m->att_in, T, M);
    VOX_TRY(conv_gemm(st, w.asp_tdnn, m->att_in, nullptr, nullptr, 1, T, 0, off0, m->att_h, nullptr, nullptr, 0));
    enc_act(st, m->att_h, (size_t)T * c.att_channels, 3);
    VOX_TRY(conv_gemm(st, w.asp_conv, m->att_h, nullptr, nullptr, 1, T, 0, off0, m->logits, nullptr, nullptr, 0));
    hipLaunchKernelGGL(k_enc_colstats, dim3((M + 63) / 64), dim3(256), 0, st, m->x, T, M, m->logits, pooled, pooled + M);
    VOX_TRY(conv_gemm(st, w.fc, pooled, nullptr, nullptr, 1, 1, 0, off0, emb_out, nullptr, nullptr, 0));
    g_conv_planes = saved_planes; g_conv_skinny_rows = saved_skinny;
    VOX_HIP(hipGetLastError());
    return VOX_OK;
}

}  // extern "C"

// ---- speech-tokenizer encoder: SEANet encoder + sliding-window transformer + downsample + split RVQ encode ----
// (tokenizer/qwen3_codec.py:1669-1773 over transformers' MimiModel; see include/voxhip.h).  Layout: fp32 time-major [t][C].
// A stride-r conv (kernel 2r, causal left pad r, zero right pad to a whole stride) is a two-tap conv_gemm over the same buffer
// viewed as [ceil(L/r)][r C]: output o reads view rows o-1 (kernel taps 0..r-1) and o (taps r..2r-1); row -1 reads as zero.
__global__ __launch_bounds__(256) void k_cenc_conv_in(const float* audio, int N, const float* w, const float* b, float* out, int nf, int k) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * nf) return;
    const int t = (int)(i / nf), co = (int)(i % nf);
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
        const int s = t - (k - 1) + j;
        if (s >= 0) acc += w[co * k + j] * audio[s];
    }
    out[i] = acc + b[co];
}
// rotate-half RoPE in place on the q and k thirds of qkv [T][3 A], A = H D
__global__ __launch_bounds__(256) void k_cenc_rope(float* qkv, int T, int H, int D, const float* inv_freq) {
    const int half = D / 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * 2 * H * half) return;
    const int d = (int)(i % half), h = (int)((i / half) % H), which = (int)((i / ((size_t)half * H)) % 2), t = (int)(i / ((size_t)half * H * 2));
    const float ang = (float)t * inv_freq[d], cs = cosf(ang), sn = sinf(ang);
    float* p = qkv + (size_t)t * 3 * H * D + (size_t)which * H * D + (size_t)h * D;
    const float x1 = p[d], x2 = p[d + half];
    p[d] = x1 * cs - x2 * sn;
    p[d + half] = x2 * cs + x1 * sn;
}
// causal attention over the last `window` keys; one wave per query row, lane = head dim (D <= 64); online softmax
__global__ __launch_bounds__(256) void k_cenc_attn(const float* qkv, float* out, int T, int H, int D, int window, float scale) {
    const int h = blockIdx.x, i = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= T) return;
    const int A = H * D, ld = 3 * A;
    const bool on = lane < D;
    const float q = on ? qkv[(size_t)i * ld + h * D + lane] * scale : 0.f;
    float mx = -INFINITY, l = 0.f, acc = 0.f;
    const int j0 = i - window + 1 > 0 ? i - window + 1 : 0;
    for (int j = j0; j <= i; ++j) {
        float s = on ? q * qkv[(size_t)j * ld + A + h * D + lane] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float nm = fmaxf(mx, s), c = expf(mx - nm), p = expf(s - nm);
        l = l * c + p;
        acc = acc * c + (on ? p * qkv[(size_t)j * ld + 2 * A + h * D + lane] : 0.f);
        mx = nm;
    }
    if (on) out[(size_t)i * A + h * D + lane] = acc / l;
}
// frame pairs for the stride-2 downsample conv (kernel 4, replicate padding): dst [To][2 H] = [x[2o] | x[2o+1]] (the last frame
// repeated when T is odd); hist [2 H] = [x[0] | x[0]] (the replicate left padding, read as the conv's look-back row)
__global__ __launch_bounds__(256) void k_cenc_pairs(const float* x, int T, int Hd, float* dst, float* hist) {
    const int To = (T + 1) / 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)(To + 1) * 2 * Hd) return;
    const int o = (int)(i / (2 * Hd)), c = (int)(i % (2 * Hd));
    if (o == To) { hist[c] = x[c % Hd]; return; }
    int t = 2 * o + c / Hd;
    t = t < T ? t : T - 1;
    dst[i] = x[(size_t)t * Hd + c % Hd];
}
// residual vector quantisation of one frame per block (MimiResidualVectorQuantizer.encode): per layer the nearest centroid by squared
// distance (lowest index on an exact tie), residual -= centroid.  One wave per centroid at a time, lanes over the dimension.
__global__ __launch_bounds__(256) void k_cenc_rvq(const float* x, const float* emb, int n_layers, int bins, int dim, int32_t* codes, int ld_codes) {
    __shared__ float r[512];
    __shared__ float bd[4];
    __shared__ int bi[4];
    const int t = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int d = threadIdx.x; d < dim; d += 256) r[d] = x[(size_t)t * dim + d];
    __syncthreads();
    for (int q = 0; q < n_layers; ++q) {
        const float* E = emb + (size_t)q * bins * dim;
        float best = INFINITY;
        int besti = 0x7fffffff;
        // eight centroids per pass: their rows' loads are independent and in flight together (the search is load-latency bound)
        for (int c0 = wave; c0 < bins; c0 += 32) {
            float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int d = lane; d < dim; d += 64) {
                const float rd = r[d];
                float e8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + 4 * u;
                    e8[u] = E[(size_t)(c < bins ? c : bins - 1) * dim + d];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const float df = rd - e8[u]; s8[u] += df * df; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float sv = s8[u];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o);
                const int c = c0 + 4 * u;
                if (c < bins && sv < best) { best = sv; besti = c; }
            }
        }
        if (lane == 0) { bd[wave] = best; bi[wave] = besti; }
        __syncthreads();
        float b = bd[0];
        int k = bi[0];
        for (int w = 1; w < 4; ++w)
            if (bd[w] < b || (bd[w] == b && bi[w] < k)) { b = bd[w]; k = bi[w]; }
        if (threadIdx.x == 0) codes[(size_t)t * ld_codes + q] = k;
        __syncthreads();
        for (int d = threadIdx.x; d < dim; d += 256) r[d] -= E[(size_t)k * dim + d];
        __syncthreads();
    }
}

struct vox_codecenc {
    vox_ctx* ctx;
    vox_codecenc_config cfg;
    vox_codecenc_weights w;
    int max_samples, hop25, max_T25;
    size_t big;
    float *A = nullptr, *B = nullptr, *C = nullptr, *tr[4] = {nullptr, nullptr, nullptr, nullptr}, *hist = nullptr, *proj = nullptr;
    int32_t* zero_slot = nullptr;
};

extern "C" {

void vox_codecenc_destroy(vox_codecenc* m) {
    if (!m) return;
    (void)hipFree(m->A); (void)hipFree(m->B); (void)hipFree(m->C); (void)hipFree(m->hist); (void)hipFree(m->proj); (void)hipFree(m->zero_slot);
    for (float* p : m->tr) (void)hipFree(p);
    delete m;
}

int vox_codecenc_create(vox_ctx* ctx, const vox_codecenc_config* cfg, const vox_codecenc_weights* w, int max_samples, vox_codecenc** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "codecenc_create: NULL");
    const vox_codecenc_config& c = *cfg;
    if (c.num_filters % (32 * c.compress) || c.compress < 1 || c.hidden % 32 || c.head_dim > 64 || c.head_dim % 2 || c.num_layers > 16 ||
        c.num_heads * c.head_dim % 32 || c.ffn % 32 || c.codebook_dim % 32 || c.codebook_dim > 512 || c.residual_kernel_size != 3 ||
        c.last_kernel_size != 3 || c.kernel_size > 16 || c.n_semantic < 1 || c.n_acoustic < 0 || max_samples < 1)
        return vox_fail(VOX_ERR_INVALID, "codecenc_create: bad config");
    vox_codecenc* m = new vox_codecenc();
    m->ctx = ctx; m->cfg = c; m->w = *w; m->max_samples = max_samples;
    // rows x channels shrinks by 2/r per stage, so the first stage bounds every SEANet buffer; + one padded stride of slack per stage
    size_t L = max_samples, ch = c.num_filters, big = (L + 64) * ch;
    m->hop25 = 1;
    for (int s = 0; s < 4; ++s) {
        if (c.ratios[s] < 1) { delete m; return vox_fail(VOX_ERR_INVALID, "codecenc_create: bad ratio"); }
        L = (L + c.ratios[s] - 1) / c.ratios[s];
        ch *= 2;
        m->hop25 *= c.ratios[s];
        big = big > (L + 64) * ch ? big : (L + 64) * ch;
    }
    m->max_T25 = (int)L;
    m->big = big;
    const size_t T = L + 2, A3 = (size_t)3 * c.num_heads * c.head_dim, wide = A3 > (size_t)c.ffn ? A3 : (size_t)c.ffn;
    bool ok = hipMalloc((void**)&m->A, big * 4) == hipSuccess && hipMalloc((void**)&m->B, big * 4) == hipSuccess &&
              hipMalloc((void**)&m->C, big * 4) == hipSuccess && hipMalloc((void**)&m->hist, (size_t)2 * c.hidden * 4) == hipSuccess &&
              hipMalloc((void**)&m->proj, T * c.codebook_dim * 4) == hipSuccess && hipMalloc((void**)&m->zero_slot, 4) == hipSuccess;
    for (int i = 0; i < 4 && ok; ++i) ok = hipMalloc((void**)&m->tr[i], T * (i < 2 ? (size_t)c.hidden * 2 : wide) * 4) == hipSuccess;
    if (!ok) { vox_codecenc_destroy(m); return vox_fail(VOX_ERR_NOMEM, "codecenc_create: hipMalloc failed"); }
    (void)hipMemset(m->zero_slot, 0, 4);
    *out = m;
    return VOX_OK;
}

int vox_codecenc_encode(vox_codecenc* m, void* stream, const float* audio, int n_samples, int32_t* codes, int32_t* n_frames, float* latents_out) {
    if (!m || !audio || !codes) return vox_fail(VOX_ERR_INVALID, "codecenc_encode: NULL");
    if (n_samples < 1 || n_samples > m->max_samples) return vox_fail(VOX_ERR_INVALID, "codecenc_encode: %d samples (capacity %d)", n_samples, m->max_samples);
    const vox_codecenc_config& c = m->cfg;
    const vox_codecenc_weights& w = m->w;
    hipStream_t st = (hipStream_t)stream;
    const int saved_planes = g_conv_planes, saved_skinny = g_conv_skinny_rows;
    g_conv_planes = 3; g_conv_skinny_rows = 48; g_conv_rows_gemm = false;
    static const int off0[1] = {0}, off3[3] = {2, 1, 0}, off2[2] = {1, 0};
    float *x = m->A, *t1 = m->B, *t2 = m->C;
    int L = n_samples, ch = c.num_filters;
    hipLaunchKernelGGL(k_cenc_conv_in, dim3(enc_grid((size_t)L * ch)), dim3(256), 0, st, audio, L, w.in_w, w.in_b, x, ch, c.kernel_size);
    for (int s = 0; s < 4; ++s) {
        const vox_codecenc_stage_w& sw = w.stage[s];
        const int r = c.ratios[s], Lo = (L + r - 1) / r;
        elu(st, x, t1, (size_t)L * ch);
        VOX_TRY(conv_gemm(st, sw.conv1, t1, nullptr, nullptr, 1, L, 0, off3, t2, nullptr, nullptr, 0));            // [L][ch / compress]
        elu(st, t2, t2, (size_t)L * (ch / c.compress));
        VOX_TRY(conv_gemm(st, sw.conv2, t2, nullptr, nullptr, 1, L, 0, off0, x, x, nullptr, 0));                   // x += conv2(...)
        elu(st, x, t1, (size_t)L * ch);
        if (Lo * r > L) VOX_HIP(hipMemsetAsync(t1 + (size_t)L * ch, 0, (size_t)(Lo * r - L) * ch * 4, st));      // zero right padding
        VOX_TRY(conv_gemm(st, sw.down, t1, nullptr, nullptr, 1, Lo, 0, off2, t2, nullptr, nullptr, 0));            // [Lo][2 ch]
        float* sw_ = x; x = t2; t2 = sw_;
        L = Lo; ch *= 2;
    }
    const int T = L, Hd = c.hidden, nh = c.num_heads, D = c.head_dim;
    float *h = m->tr[0], *nrm = m->tr[1], *qkv = m->tr[2], *att = m->tr[3];
    elu(st, x, t1, (size_t)T * ch);
    VOX_TRY(conv_gemm(st, w.last, t1, nullptr, nullptr, 1, T, 0, off3, h, nullptr, nullptr, 0));                   // h [T][hidden]
    for (int l = 0; l < c.num_layers; ++l) {
        const vox_mimi_layer_w& lw = w.layers[l];
        hipLaunchKernelGGL(k_layernorm_f32, dim3(T), dim3(256), 0, st, h, lw.ln1_w, lw.ln1_b, nrm, Hd, c.ln_eps);
        VOX_TRY(conv_gemm(st, lw.qkv, nrm, nullptr, nullptr, 1, T, 0, off0, qkv, nullptr, nullptr, 0));            // [T][3A]
        hipLaunchKernelGGL(k_cenc_rope, dim3(enc_grid((size_t)T * nh * D)), dim3(256), 0, st, qkv, T, nh, D, w.inv_freq);
        hipLaunchKernelGGL(k_cenc_attn, dim3(nh, (T + 3) / 4), dim3(256), 0, st, qkv, att, T, nh, D, c.window, 1.0f / sqrtf((float)D));
        VOX_TRY(conv_gemm(st, lw.o, att, nullptr, nullptr, 1, T, 0, off0, h, h, lw.scale1, 0));
        hipLaunchKernelGGL(k_layernorm_f32, dim3(T), dim3(256), 0, st, h, lw.ln2_w, lw.ln2_b, nrm, Hd, c.ln_eps);
        VOX_TRY(conv_gemm(st, lw.fc1, nrm, nullptr, nullptr, 1, T, 0, off0, qkv, nullptr, nullptr, 1));            // GELU
        VOX_TRY(conv_gemm(st, lw.fc2, qkv, nullptr, nullptr, 1, T, 0, off0, h, h, lw.scale2, 0));
    }
    const int To = (T + 1) / 2;
    hipLaunchKernelGGL(k_cenc_pairs, dim3(enc_grid((size_t)(To + 1) * 2 * Hd)), dim3(256), 0, st, h, T, Hd, nrm, m->hist);
    float* lat = latents_out ? latents_out : att;
    VOX_TRY(conv_gemm(st, w.downsample, nrm, m->hist, m->zero_slot, 1, To, 1, off2, lat, nullptr, nullptr, 0));    // [To][hidden]
    const int nq = c.n_semantic + c.n_acoustic;
    VOX_TRY(conv_gemm(st, w.sem_proj, lat, nullptr, nullptr, 1, To, 0, off0, m->proj, nullptr, nullptr, 0));
    hipLaunchKernelGGL(k_cenc_rvq, dim3(To), dim3(256), 0, st, m->proj, w.sem_emb, c.n_semantic, c.codebook_size, c.codebook_dim, codes, nq);
    if (c.n_acoustic > 0) {
        VOX_TRY(conv_gemm(st, w.ac_proj, lat, nullptr, nullptr, 1, To, 0, off0, m->proj, nullptr, nullptr, 0));
        hipLaunchKernelGGL(k_cenc_rvq, dim3(To), dim3(256), 0, st, m->proj, w.ac_emb, c.n_acoustic, c.codebook_size, c.codebook_dim,
                           codes + c.n_semantic, nq);
    }
    if (n_frames) *n_frames = To;
    g_conv_planes = saved_planes; g_conv_skinny_rows = saved_skinny;
    VOX_HIP(hipGetLastError());
    return VOX_OK;
}

}  // extern "C"
