// gfx950 kernels of the speech-LM decode step: weight-streaming linear (fixed-order GEMV), RMSNorm,
// RoPE (+ per-head q/k norm + paged KV append), chunked paged attention.  wave64, 256-thread blocks.
//
// Weight-streaming linear: decode at B<=8 rows is HBM-bound (1 FLOP/byte/row) — weights go straight
// from HBM to VGPRs with 16-byte non-temporal loads (no LDS round trip: each weight byte is used by
// exactly one wave), the few activation rows sit in LDS, MFMA is deliberately not used here.
#include "vox_internal.h"

// ================================================================================================
// linear
// ================================================================================================
enum { PRO_COPY = 0, PRO_RMSNORM = 1, PRO_ATTN = 2 };
enum { EPI_STORE = 0, EPI_SILU = 1, EPI_SILU_MUL = 2 };

struct LinArgs {
    const bf16_t *W, *W2, *bias, *x, *residual, *nw;
    bf16_t *y, *x_out;
    const float *part_o, *part_ml;
    const int* kvlen;
    const int* x_rows;   // optional row indirection: x row b = x + x_rows[b]*x_stride
    long x_stride, x_out_stride;
    float eps;
    int B, N, K, Hq, D, max_chunks;
    const bf16_t* post_nw;   // optional: RMSNorm weight of the NEXT linear; post_out = norm(y) written beside y (17+ rows path)
    bf16_t* post_out;
    float post_eps;
    int keep;            // 1: weights are re-read soon (depth loop): plain loads, let them live in the Infinity Cache
    const bf16_t *W_frag, *W2_frag, *x_frag;   // fragment-major operands of the 9..32 rows path (see LinearCall)
    bf16_t* y_frag;
    int y_rowmajor;
    int row_tiles;       // k_gemv only: > 1 = that many row tiles of BT rows, one block per (column group, row tile)
    int rt_rows;         // k_gemm_fullk<.., ROWSPLIT>: rows per row tile (0 = 16 * MT); 8 or 4: SUB-TILE row split for <= 16 rows (below)
    int no_one_seg;      // k_linear_mfma A/B knob (VOX_MFMA_ONESEG=0): row statistics by the separate pass even when K fits one segment
};

#ifndef VOX_XFIRST
#define VOX_XFIRST 2
#endif
#define VOX_CONST_AS __attribute__((address_space(4)))
#ifndef VOX_DS_PICK_LATE
#define VOX_DS_PICK_LATE 0      // 1: the step input (pick) behind layer 0's first weight requests — measured: no difference, six spilled registers
#endif
#ifndef VOX_GRAN_ASM
#define VOX_GRAN_ASM 4    // persistent kernels: a poll pass's granule requests written as asm (see gran_poll_pass / gran_poll_all); 0: atomic loads
#endif
#ifndef VOX_KV_ASM
#define VOX_KV_ASM 1      // decode attention: K/V tile requests written as asm (see k_attn_decode8)
#endif
#ifndef VOX_MLP_D_EARLY
#define VOX_MLP_D_EARLY 1
#endif
#ifndef VOX_MLP_C2_LATE
#define VOX_MLP_C2_LATE 0
#endif
#ifdef VOX_DEV_KNOBS
// development builds: chain trace.  Thread 0 of block 0 of every instrumented launch keeps up to 8 s_memrealtime stamps (100 MHz)
// in registers and writes one 16-word record {kind, n, t0..} when it ends (launch order = record order: the chains are
// dependent); thread 0 of the LAST block writes {entry, exit} into a second stream of records with the same numbering (how long
// the grid takes to start and to drain).  Word 0 / 1 of the buffer count the records of the two streams.  vox_dev_set_trace().
__device__ unsigned long long* g_vox_trace = nullptr;
extern "C" int vox_dev_set_trace(void* p) {
    unsigned long long* q = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_vox_trace), &q, sizeof(q)) == hipSuccess ? 0 : -1;
}
#define VOX_TRACE_CAP 6000
#define VOX_TR_DECL                                                                                                          \
    unsigned long long tr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_slot = ~0ull, tr_slot2 = ~0ull;                                \
    const bool tr_first = g_vox_trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0;                             \
    const bool tr_last = g_vox_trace && threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1;     \
    if (tr_first || tr_last) tr_t[0] = wall_clock64();                                                                       \
    if (tr_first) tr_slot = atomicAdd(g_vox_trace, 1ull);                                                                    \
    if (tr_last) tr_slot2 = atomicAdd(g_vox_trace + 1, 1ull);
#define VOX_TR(k) if (tr_first) tr_t[k] = wall_clock64();
#define VOX_TR_END(kind, n)                                                                                                  \
    if (tr_first || tr_last) {                                                                                               \
        const unsigned long long tr_e = wall_clock64();                                                                      \
        if (tr_first && tr_slot < VOX_TRACE_CAP) {                                                                           \
            unsigned long long* r = g_vox_trace + 16 * (tr_slot + 1);                                                        \
            r[0] = (unsigned long long)(kind); r[1] = (unsigned long long)(n); r[2] = ((unsigned long long)gridDim.x << 32) | gridDim.y; \
            for (int i_ = 0; i_ < 8; ++i_) r[3 + i_] = tr_t[i_];                                                              \
            r[11] = tr_e;                                                                                                    \
        }                                                                                                                    \
        if (tr_last && tr_slot2 < VOX_TRACE_CAP) {                                                                           \
            unsigned long long* r = g_vox_trace + 16 * (VOX_TRACE_CAP + 1) + 2 * tr_slot2;                                    \
            r[0] = tr_t[0]; r[1] = tr_e;                                                                                     \
        }                                                                                                                    \
    }
#else
#define VOX_TR_DECL
#define VOX_TR(k)
#define VOX_TR_END(kind, n)
#endif

// element offset of (row r, column k) of a [*, K] matrix in fragment-major form
__device__ __forceinline__ size_t frag_off(int r, int k, int K) {
    return ((size_t)(r >> 4) * (K >> 5) + (k >> 5)) * 512 + (size_t)(((r & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7));
}

__device__ __forceinline__ const uint4* x_row_ptr(const LinArgs& a, int row) {
    const long r = a.x_rows ? a.x_rows[row] : row;
    return reinterpret_cast<const uint4*>(a.x + r * a.x_stride);
}

__device__ __forceinline__ uint4 norm_chunk(uint4 v, uint4 w, float rinv) {
    uint4 o;
    o.x = pack_bf2((bflo(v.x) * rinv) * bflo(w.x), (bfhi(v.x) * rinv) * bfhi(w.x));
    o.y = pack_bf2((bflo(v.y) * rinv) * bflo(w.y), (bfhi(v.y) * rinv) * bfhi(w.y));
    o.z = pack_bf2((bflo(v.z) * rinv) * bflo(w.z), (bfhi(v.z) * rinv) * bfhi(w.z));
    o.w = pack_bf2((bflo(v.w) * rinv) * bflo(w.w), (bfhi(v.w) * rinv) * bfhi(w.w));
    return o;
}

// skip_row0: row `wave` of tile 0 was already staged from registers by the caller (fast path)
template <int PRO>
__device__ __forceinline__ void stage_x(const LinArgs& a, uint4* xs, int b0, int bt, int bt_cap, bool skip_row0 = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = a.K >> 3;
    if (PRO == PRO_COPY) {
        for (int b = 0; b < bt; ++b) {
            const uint4* src = x_row_ptr(a, b0 + b);
            for (int i = tid; i < nch; i += 256) xs[b * nch + i] = src[i];
        }
    } else if (PRO == PRO_RMSNORM) {
        const uint4* nw = reinterpret_cast<const uint4*>(a.nw);
        for (int b = wave + (skip_row0 ? 4 : 0); b < bt; b += 4) {
            const uint4* xr = x_row_ptr(a, b0 + b);
            float s = 0.0f;
            for (int c = lane; c < nch; c += 64) s = sq8(xr[c], s);
            s = butterfly<64>(s);
            const float rinv = 1.0f / sqrtf(s / (float)a.K + a.eps);
            for (int c = lane; c < nch; c += 64) {
                const uint4 o = norm_chunk(xr[c], nw[c], rinv);
                xs[b * nch + c] = o;
                if (a.x_out && blockIdx.x == 0)
                    reinterpret_cast<uint4*>(a.x_out + (size_t)(b0 + b) * a.x_out_stride)[c] = o;
            }
        }
    } else {  // PRO_ATTN: merge the attention partials of row b into x[b, h*D+d]
        // phase 1: one thread per (row, head): global max, per-chunk weights exp2((m_c-M)*log2e) and the
        // sequentially accumulated denominator -> LDS;  phase 2: one thread per element, coalesced over d.
        bf16_t* xb = reinterpret_cast<bf16_t*>(xs);
        float* wts = reinterpret_cast<float*>(xs + (size_t)bt_cap * nch);   // [bt][Hq][max_chunks+1]
        const int HD = a.Hq * a.D, ws = a.max_chunks + 1;
        for (int p = tid; p < bt * a.Hq; p += 256) {
            const int b = p / a.Hq, h = p % a.Hq;
            const int row = b0 + b;
            const int nc = (a.kvlen[row] + VOX_TC - 1) / VOX_TC;
            const float2* ml = reinterpret_cast<const float2*>(a.part_ml + ((size_t)row * a.Hq + h) * a.max_chunks * 2);
            float M = -INFINITY;
            for (int c0 = 0; c0 < nc; c0 += 8) {         // 8 independent loads in flight
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (c0 + j < nc) ? ml[c0 + j] : make_float2(-INFINITY, 0.0f);
#pragma unroll
                for (int j = 0; j < 8; ++j) M = fmaxf(M, v[j].x);
            }
            float L = 0.0f;
            for (int c0 = 0; c0 < nc; c0 += 8) {
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (c0 + j < nc) ? ml[c0 + j] : make_float2(-INFINITY, 0.0f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (c0 + j < nc) {
                        const float w = exp2_c((v[j].x - M) * VOX_LOG2E);
                        L = __fmaf_rn(v[j].y, w, L);
                        wts[(size_t)p * ws + c0 + j] = w;
                    }
                }
            }
            wts[(size_t)p * ws + a.max_chunks] = L;
        }
        __syncthreads();
        for (int e = tid; e < bt * HD; e += 256) {
            const int b = e / HD, h = (e % HD) / a.D, d = e % a.D;
            const int row = b0 + b;
            const int nc = (a.kvlen[row] + VOX_TC - 1) / VOX_TC;
            const float* po = a.part_o + ((size_t)row * a.Hq + h) * a.max_chunks * a.D + d;
            const float* w = wts + (size_t)(b * a.Hq + h) * ws;
            float O = 0.0f;
            for (int c0 = 0; c0 < nc; c0 += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (c0 + j < nc) ? po[(size_t)(c0 + j) * a.D] : 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c0 + j < nc) O = __fmaf_rn(v[j], w[c0 + j], O);
            }
            xb[e] = f2bf(O / w[a.max_chunks]);
        }
    }
}

template <int BT, int R, int PRO, int EPI>
__global__ __launch_bounds__(256) void k_linear(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* xs = reinterpret_cast<uint4*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = a.K >> 3;
    constexpr int OUT = (EPI == EPI_SILU_MUL) ? R / 2 : R;
    constexpr int U = (R * BT >= 16) ? 2 : 4;  // chunk-steps issued together (loads in flight per lane = U*R); 4 vs 2 at R*BT >= 16: no difference (measured)
    const int n0 = (blockIdx.x * 4 + wave) * OUT;

    const uint4* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int n = n0 + (EPI == EPI_SILU_MUL ? (r % OUT) : r);
        n = n < a.N ? n : a.N - 1;
        const bf16_t* base = (EPI == EPI_SILU_MUL && r >= OUT) ? a.W2 : a.W;
        wrow[r] = reinterpret_cast<const uint4*>(base + (size_t)n * a.K);
    }

    // The first group of weight chunks is requested BEFORE the activation prologue: weights do not depend on
    // x, so their HBM latency overlaps the norm / merge work and the barrier.
    // epilogue operands (residual, bias) of the first batch tile are requested up front as well: lane o*BT+b owns output
    // (o, b); reading them only in the epilogue would add one more dependent HBM/L2 round trip to every kernel.
    float res_pre = 0.0f, bias_pre = 0.0f;
    {
        const int po = lane / BT, pb = lane % BT;
        if (EPI != EPI_SILU_MUL && lane < OUT * BT && pb < a.B && n0 + po < a.N) {
            if (a.residual) res_pre = bf2f(a.residual[(size_t)pb * a.N + n0 + po]);
            if (a.bias) bias_pre = bf2f(a.bias[n0 + po]);
        }
    }
    // RMSNorm prologue fast path: the activation row this wave normalises (row `wave` of tile 0) is requested
    // first, so it is not queued behind the weight stream (vmcnt retires in order).
    constexpr int XC = 4;
    const bool xfast = (PRO == PRO_RMSNORM) && nch <= 64 * XC && wave < a.B;
    uint4 xr0[XC], nw0[XC];
    if (xfast) {
        const uint4* xr = x_row_ptr(a, wave);
        const uint4* nwp = reinterpret_cast<const uint4*>(a.nw);
#pragma unroll
        for (int j = 0; j < XC; ++j) {
            xr0[j] = make_uint4(0, 0, 0, 0);
            nw0[j] = xr0[j];
            if (lane + 64 * j < nch) {
                xr0[j] = xr[lane + 64 * j];
                nw0[j] = nwp[lane + 64 * j];
            }
        }
    }
    uint4 w[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (lane + 64 * u < nch) w[u][r] = ldg_nt(wrow[r] + lane + 64 * u);
    if (xfast) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < XC; ++j) s = sq8(xr0[j], s);   // absent chunks are zero: same sum as the guarded loop
        s = butterfly<64>(s);
        const float rinv = 1.0f / sqrtf(s / (float)a.K + a.eps);
#pragma unroll
        for (int j = 0; j < XC; ++j) {
            const int c = lane + 64 * j;
            if (c < nch) {
                const uint4 o = norm_chunk(xr0[j], nw0[j], rinv);
                xs[wave * nch + c] = o;
                if (a.x_out && blockIdx.x == 0) reinterpret_cast<uint4*>(a.x_out + (size_t)wave * a.x_out_stride)[c] = o;
            }
        }
    }

    for (int b0 = 0; b0 < a.B; b0 += BT) {
        const int bt = (a.B - b0) < BT ? (a.B - b0) : BT;
        if (b0 > 0) __syncthreads();
        stage_x<PRO>(a, xs, b0, bt, BT, (PRO == PRO_RMSNORM) && b0 == 0 && nch <= 64 * XC);
        __syncthreads();

        float acc[R][BT];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < BT; ++b) acc[r][b] = 0.0f;

        for (int c = lane; c < nch; c += 64 * U) {
            if (c != lane || b0 > 0) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (c + 64 * u < nch) w[u][r] = ldg_nt(wrow[r] + c + 64 * u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c + 64 * u < nch) {
                    if constexpr (BT == 1) {
                        const uint4 xv = xs[c + 64 * u];
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][0] = dot8(w[u][r], xv, acc[r][0]);
                    } else {
                        // two rows per v_pk_fma_f32 (same IEEE fma per half, same order as dot8: bit-identical sums, half
                        // the FMA issue slots); the unpacked weight chunk is shared by all row pairs
                        float wf[R][8];
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const uint4 wq = w[u][r];
                            wf[r][0] = bflo(wq.x); wf[r][1] = bfhi(wq.x); wf[r][2] = bflo(wq.y); wf[r][3] = bfhi(wq.y);
                            wf[r][4] = bflo(wq.z); wf[r][5] = bfhi(wq.z); wf[r][6] = bflo(wq.w); wf[r][7] = bfhi(wq.w);
                        }
#pragma unroll
                        for (int bp = 0; bp < BT / 2; ++bp) {
                            const uint4 x0 = xs[(2 * bp) * nch + c + 64 * u], x1 = xs[(2 * bp + 1) * nch + c + 64 * u];
                            const vox_f2 xp[8] = {{bflo(x0.x), bflo(x1.x)}, {bfhi(x0.x), bfhi(x1.x)}, {bflo(x0.y), bflo(x1.y)},
                                                  {bfhi(x0.y), bfhi(x1.y)}, {bflo(x0.z), bflo(x1.z)}, {bfhi(x0.z), bfhi(x1.z)},
                                                  {bflo(x0.w), bflo(x1.w)}, {bfhi(x0.w), bfhi(x1.w)}};
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                vox_f2 a2 = {acc[r][2 * bp], acc[r][2 * bp + 1]};
#pragma unroll
                                for (int e = 0; e < 8; ++e) a2 = __builtin_elementwise_fma((vox_f2){wf[r][e], wf[r][e]}, xp[e], a2);
                                acc[r][2 * bp] = a2.x;
                                acc[r][2 * bp + 1] = a2.y;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < BT; ++b) acc[r][b] = butterfly<64>(acc[r][b]);

#pragma unroll
        for (int o = 0; o < OUT; ++o)
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                if (lane == o * BT + b && b < bt && n0 + o < a.N) {
                    const int n = n0 + o;
                    const size_t oi = (size_t)(b0 + b) * a.N + n;
                    bf16_t r;
                    if (EPI == EPI_SILU_MUL) {
                        const float g = bfround(acc[o][b]);
                        const float u = bfround(acc[o + OUT][b]);
                        r = f2bf(bfround(silu_c(g)) * u);
                    } else {
                        float v = acc[o][b];
                        if (a.bias) v = v + (b0 == 0 ? bias_pre : bf2f(a.bias[n]));
                        r = f2bf(v);
                        if (EPI == EPI_SILU) r = f2bf(silu_c(bf2f(r)));
                        if (a.residual) r = f2bf((b0 == 0 ? res_pre : bf2f(a.residual[oi])) + bf2f(r));
                    }
                    a.y[oi] = r;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Register-resident GEMV for 1..2 rows: K == 512*KC, every wave is independent (no LDS, no barrier).
// Lane l owns chunks l, l+64, ... of every row it touches — exactly the canonical DOT / RMSNorm lane
// assignment — so the activation rows go straight from L2 into the registers that feed dot8, every wave
// redoes the (tiny) RMSNorm reduction itself instead of waiting on wave 0 behind a barrier, and all of a
// wave's weight bytes (R rows x K) are requested in one burst before anything waits.  Same arithmetic,
// bit for bit, as k_linear.
template <int BT, int KC, int R, int PRO, int EPI>
__global__ __launch_bounds__(256) void k_gemv(LinArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int OUT = (EPI == EPI_SILU_MUL) ? R / 2 : R;
    // 3..8 rows: the same kernel per row tile.  Block id = (x / 8) * 8T + tile * 8 + x % 8: the T blocks that stream the same
    // weight rows are 8 ids apart = same XCD, dispatched together, so the weights cross HBM once and the other T - 1 reads hit
    // that XCD's L2.  Per-row arithmetic is unchanged (bit-exact with the single-tile launch).
    int bx = blockIdx.x, rb = 0;
    if (a.row_tiles > 1) {
        bx = (blockIdx.x / (8 * a.row_tiles)) * 8 + (blockIdx.x & 7);
        rb = ((blockIdx.x >> 3) % a.row_tiles) * BT;
    }
    const int n0 = (bx * 4 + wave) * OUT;
    if (n0 >= a.N) return;
    const int nb = a.B - rb;      // rows of this tile (>= 1)

    uint4 xv[BT][KC], nwv[KC];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        const uint4* xr = x_row_ptr(a, rb + (b < nb ? b : 0));
#pragma unroll
        for (int j = 0; j < KC; ++j) xv[b][j] = xr[lane + 64 * j];
    }
    if (PRO == PRO_RMSNORM) {
        const uint4* nwp = reinterpret_cast<const uint4*>(a.nw);
#pragma unroll
        for (int j = 0; j < KC; ++j) nwv[j] = nwp[lane + 64 * j];
    }
    uint4 w[R][KC];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int n = n0 + (EPI == EPI_SILU_MUL ? (r % OUT) : r);
        n = n < a.N ? n : a.N - 1;
        const bf16_t* base = (EPI == EPI_SILU_MUL && r >= OUT) ? a.W2 : a.W;
        const uint4* wr = reinterpret_cast<const uint4*>(base + (size_t)n * a.K);
        if (a.keep) {
#pragma unroll
            for (int j = 0; j < KC; ++j) w[r][j] = wr[lane + 64 * j];
        } else {
#pragma unroll
            for (int j = 0; j < KC; ++j) w[r][j] = ldg_nt(wr + lane + 64 * j);
        }
    }
    float res_pre = 0.0f, bias_pre = 0.0f;
    {
        const int po = lane / BT, pb = lane % BT;
        if (EPI != EPI_SILU_MUL && lane < OUT * BT && pb < nb && n0 + po < a.N) {
            if (a.residual) res_pre = bf2f(a.residual[(size_t)(rb + pb) * a.N + n0 + po]);
            if (a.bias) bias_pre = bf2f(a.bias[n0 + po]);
        }
    }
    if (PRO == PRO_RMSNORM) {
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < KC; ++j) s = sq8(xv[b][j], s);
            s = butterfly<64>(s);
            const float rinv = 1.0f / sqrtf(s / (float)a.K + a.eps);
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                xv[b][j] = norm_chunk(xv[b][j], nwv[j], rinv);
                if (a.x_out && bx == 0 && wave == 0 && b < nb)
                    reinterpret_cast<uint4*>(a.x_out + (size_t)(rb + b) * a.x_out_stride)[lane + 64 * j] = xv[b][j];
            }
        }
    }
    float acc[R][BT];
    if constexpr (BT == 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < KC; ++j) s = dot8(w[r][j], xv[0][j], s);
            acc[r][0] = butterfly<64>(s);
        }
    } else {
        // two rows per v_pk_fma_f32: (w,w) x (x_b, x_b+1) + (acc_b, acc_b+1) — each half is the same IEEE fma, in the
        // same e / chunk order as dot8, so the sums are bit-identical while the FMA issue count halves; the unpacked
        // activation pair is shared by the R weight rows of the wave
        constexpr int BP = BT >= 2 ? BT / 2 : 1;   // (BT == 1 never takes this branch)
        vox_f2 acc2[R][BP];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int bp = 0; bp < BP; ++bp) acc2[r][bp] = (vox_f2){0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            vox_f2 xp[BP][8];
#pragma unroll
            for (int bp = 0; bp < BP; ++bp) {
                const uint4 x0 = xv[2 * bp][j], x1 = xv[2 * bp + 1][j];
                xp[bp][0] = (vox_f2){bflo(x0.x), bflo(x1.x)}; xp[bp][1] = (vox_f2){bfhi(x0.x), bfhi(x1.x)};
                xp[bp][2] = (vox_f2){bflo(x0.y), bflo(x1.y)}; xp[bp][3] = (vox_f2){bfhi(x0.y), bfhi(x1.y)};
                xp[bp][4] = (vox_f2){bflo(x0.z), bflo(x1.z)}; xp[bp][5] = (vox_f2){bfhi(x0.z), bfhi(x1.z)};
                xp[bp][6] = (vox_f2){bflo(x0.w), bflo(x1.w)}; xp[bp][7] = (vox_f2){bfhi(x0.w), bfhi(x1.w)};
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint4 wq = w[r][j];
                const float wf[8] = {bflo(wq.x), bfhi(wq.x), bflo(wq.y), bfhi(wq.y), bflo(wq.z), bfhi(wq.z), bflo(wq.w), bfhi(wq.w)};
#pragma unroll
                for (int bp = 0; bp < BP; ++bp)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        acc2[r][bp] = __builtin_elementwise_fma((vox_f2){wf[e], wf[e]}, xp[bp][e], acc2[r][bp]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int bp = 0; bp < BP; ++bp) {
                acc[r][2 * bp] = butterfly<64>(acc2[r][bp].x);
                acc[r][2 * bp + 1] = butterfly<64>(acc2[r][bp].y);
            }
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o)
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            if (lane == o * BT + b && b < nb && n0 + o < a.N) {
                const int n = n0 + o;
                const size_t oi = (size_t)(rb + b) * a.N + n;
                bf16_t r;
                if (EPI == EPI_SILU_MUL) {
                    const float g = bfround(acc[o][b]);
                    const float u = bfround(acc[o + OUT][b]);
                    r = f2bf(bfround(silu_c(g)) * u);
                } else {
                    float v = acc[o][b];
                    if (a.bias) v = v + bias_pre;
                    r = f2bf(v);
                    if (EPI == EPI_SILU) r = f2bf(silu_c(bf2f(r)));
                    if (a.residual) r = f2bf(res_pre + bf2f(r));
                }
                a.y[oi] = r;
            }
        }
}

#ifndef VOX_GEMV_COLS      // output columns per wave (development knob; 0 = heuristic)
#define VOX_GEMV_COLS 0
#endif
template <int BT, int KC, int PRO, int EPI>
static int launch_gemv_k(hipStream_t st, const LinArgs& a) {
    constexpr bool SM = (EPI == EPI_SILU_MUL);
    // columns per wave: enough waves to cover the 1024 SIMDs a few times over, as many bytes in flight per wave
    // as the register file allows (R*KC 16-byte loads per lane)
    int cols = VOX_GEMV_COLS;
    if (cols == 0) cols = (a.N / (SM ? 1 : 1) >= 4096 && KC <= 4) ? 2 : 1;
    if (SM) cols = cols > 2 ? 2 : cols;
    while (cols > 1 && cols * (SM ? 2 : 1) * KC > 24) cols >>= 1;
    if (BT >= 4 && !SM && a.N >= 2048) cols = 2;          // share the unpacked activation pairs between two weight rows
    if (BT >= 4) while (cols > 1 && cols * (SM ? 2 : 1) * KC > 8) cols >>= 1;
#define VOX_GV(C_)                                                                                       \
    if (cols == C_) {                                                                                    \
        const int gx = (a.N + 4 * C_ - 1) / (4 * C_);                                                    \
        if (a.row_tiles > 1 && gx % 8) return vox_fail(VOX_ERR_INVALID, "gemv: row tiles need N/(4 cols) %% 8 == 0"); \
        hipLaunchKernelGGL((k_gemv<BT, KC, (SM ? 2 * C_ : C_), PRO, EPI>), dim3(gx * (a.row_tiles > 1 ? a.row_tiles : 1)), \
                           dim3(256), 0, st, a);                                                         \
        return VOX_OK;                                                                                   \
    }
    VOX_GV(1) VOX_GV(2) VOX_GV(4)
#undef VOX_GV
    return vox_fail(VOX_ERR_INVALID, "gemv: no variant");
}
template <int PRO, int EPI>
static int launch_gemv(hipStream_t st, const LinArgs& a, bool* handled) {
    *handled = true;
    const int kc = a.K / 512;
    const int rows_per_block = a.row_tiles > 1 ? 2 : a.B;
#define VOX_GK(B_, K_) if (rows_per_block <= B_ && kc == K_) return launch_gemv_k<B_, K_, PRO, EPI>(st, a);
    VOX_GK(1, 2) VOX_GK(1, 4) VOX_GK(1, 6) VOX_GK(1, 8) VOX_GK(1, 12) VOX_GK(1, 16)
    VOX_GK(2, 2) VOX_GK(2, 4) VOX_GK(2, 6) VOX_GK(2, 8) VOX_GK(2, 12)
#undef VOX_GK
    *handled = false;
    return VOX_OK;
}

template <int BT, int R, int PRO, int EPI>
static int launch_linear_t(hipStream_t st, const LinArgs& a) {
    constexpr int OUT = (EPI == EPI_SILU_MUL) ? R / 2 : R;
    const int grid = (a.N + 4 * OUT - 1) / (4 * OUT);
    const size_t smem = (size_t)BT * a.K * 2 + (PRO == PRO_ATTN ? (size_t)BT * a.Hq * (a.max_chunks + 1) * 4 : 0);
    auto kern = k_linear<BT, R, PRO, EPI>;
    if (smem > 64 * 1024) {
        static bool done = false;  // per instantiation
        if (!done) {
            VOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, a);
    return VOX_OK;
}

static bool gemv_row_tiles_ok(const LinArgs& a, int pro) {
    const int kc = a.K / 512;
    if (!(kc == 2 || kc == 4 || kc == 6 || kc == 8 || kc == 12)) return false;
#ifdef VOX_DEV_KNOBS
    static int mode = -1;       // development builds: 0 never, 1 cache-resident weights only, 2 always, 3 shipped rule
    if (mode < 0) { const char* e = getenv("VOX_ROW_TILES"); mode = e ? atoi(e) : 3; }
    if (mode == 0 || (mode == 1 && !a.keep)) return false;
    if (mode == 1 || mode == 2) return true;
#endif
    // measured (B = 4 / 6 / 7 / 8 frames): always a win for the cache-resident depth weights and for copy-prologue linears
    // (o_proj, down); with 4 row tiles of weights streamed from HBM the norm-prologue linears (every wave redoes the norm of
    // its two rows) are faster in the LDS-staged 8-row kernel
    return a.keep || a.B <= 6 || pro == PRO_COPY;
}
template <int PRO, int EPI>
static int launch_linear_pe(hipStream_t st, const LinArgs& a, int n_cu) {
    // batch tile: smallest of {1,2,4,8} covering B (B>8 loops tiles of 8 inside the kernel)
#ifndef VOX_NO_GEMV
    if (PRO != PRO_ATTN && a.B <= 2 && a.K % 512 == 0) {       // register-resident, barrier-free variant
        bool handled = false;
        const int rc = launch_gemv<(PRO == PRO_ATTN ? PRO_COPY : PRO), EPI>(st, a, &handled);
        if (handled) return rc;
    }
    if (PRO != PRO_ATTN && a.B >= 3 && a.B <= 8 && a.K % 512 == 0 && a.N % 64 == 0 && gemv_row_tiles_ok(a, PRO)) {
        LinArgs t = a;                                          // 3..8 rows: the 2-row kernel per row pair
        t.row_tiles = (a.B + 1) / 2;
        bool handled = false;
        const int rc = launch_gemv<(PRO == PRO_ATTN ? PRO_COPY : PRO), EPI>(st, t, &handled);
        if (handled) return rc;
    }
#endif
    int bt = a.B <= 1 ? 1 : a.B <= 2 ? 2 : a.B <= 4 ? 4 : 8;
    while (bt > 1 && (size_t)bt * a.K * 2 > 144 * 1024) bt >>= 1;
    // rows per wave: keep >= ~1 block per CU; fewer rows/wave when N is small
    constexpr bool SM = (EPI == EPI_SILU_MUL);
    const int outs = a.N;
    int r;  // outputs per wave: keep >= 2 blocks per CU in flight where N allows (latency hiding)
    (void)n_cu;
    if (SM) r = (bt >= 4 && outs >= 8192) ? 2 : 1;   // one gate row + one up row per wave (two of each for very wide FFNs at >= 4 rows:
                                                     // GLM-4-Voice B=8 token 12.7 -> 11.3 ms)
    else r = ((bt <= 2 && outs >= 8192) || (bt >= 4 && outs >= 2048)) ? 2 : 1;   // >= 4 rows: two weight rows share the
                                                                                     // unpacked activation pairs
    if ((size_t)bt * a.K * 2 > 160 * 1024) return vox_fail(VOX_ERR_INVALID, "linear: K too large for LDS staging");
#define VOX_LIN(BT_, R_)                                                             \
    if (bt == BT_ && r == R_) return launch_linear_t<BT_, (SM ? 2 * R_ : R_), PRO, EPI>(st, a);
    VOX_LIN(1, 4) VOX_LIN(1, 2) VOX_LIN(1, 1) VOX_LIN(2, 4) VOX_LIN(2, 2) VOX_LIN(2, 1)
    VOX_LIN(4, 2) VOX_LIN(4, 1) VOX_LIN(8, 2) VOX_LIN(8, 1)
#undef VOX_LIN
    return vox_fail(VOX_ERR_INVALID, "linear: no kernel variant");
}

// ================================================================================================
// linear for 9..N rows: bf16 MFMA, weights streamed once per 16/32-row tile.
// Used for batched decode (B > 8) and prefill.  Each wave owns 16 output columns (weight rows): the B
// fragment of v_mfma_f32_16x16x32_bf16 is exactly one 16-byte load per lane straight from HBM
// (row n0+(lane&15), k-block (lane>>4)*8), the activation tile sits in LDS (row-padded by 16 B so the 16
// rows of an A fragment hit distinct banks).  fp32 accumulation in MFMA order: parity with the oracle is to
// bf16 rounding, not bit-exact (rows <= 8 keep the fixed-order VALU kernel above).
// ================================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define MF_KS 1024   // K segment staged in LDS at a time

__device__ __forceinline__ bf16x8_t as_bf8(uint4 v) {
    union { uint4 u; bf16x8_t b; } c;
    c.u = v;
    return c.b;
}

// FULL: K % MF_KS == 0 and N % 16 == 0 — every range guard compiles away (at one wave per SIMD the kernel is
// instruction-issue bound, each guard is a divergent-branch sequence); padded batch rows re-read the last real row.
// DEPTH > 1: that many weight segments in flight (a ring of DEPTH register buffers, the loop unrolled DEPTH times so that every buffer
// index is static): iteration s requests the activations of s + 1 and then the weights of s + DEPTH.  vmcnt retires in order, so the
// wait for x(s + 1) also waits for every weight segment requested before it, i.e. w(j) is complete DEPTH - 1 iterations after its
// request: with DEPTH = 3 and 1024-wide segments a segment has ~2 iterations (about the HBM latency) to arrive, 96 KB per CU in flight.
// Measured on the 13 696-wide GLM down projection: slower than one 2048-wide segment in flight (see launch_linear_mfma_pe); opt-in.
// Which k-steps a wave multiplies and in which order is unchanged: bit-identical.
template <int B_> struct vox_ic { static constexpr int value = B_; };
template <int MT, int PRO, int EPI, bool FULL, int KSEG, int DEPTH = 1>
__global__ __launch_bounds__(256) void k_linear_mfma(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BT = 16 * MT;
    constexpr int LDK = KSEG + 8;                  // padded row stride (elements)
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);   // [BT][LDK]
    float* rinv = reinterpret_cast<float*>(smem + (size_t)BT * LDK * 2);   // [BT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // one 16-column tile per block; the four waves interleave the K steps (step s -> wave s%4) and their partial
    // accumulators are summed in wave order through LDS: N/16 blocks keep the whole chip streaming even at N=1024.
    const int n0 = blockIdx.x * 16;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    int nrow = n0 + fr;
    nrow = nrow < a.N ? nrow : a.N - 1;
    const bf16_t* wrow = a.W + (size_t)nrow * a.K;
    const bf16_t* wrow2 = (EPI == EPI_SILU_MUL) ? a.W2 + (size_t)nrow * a.K : nullptr;

    for (int b0 = 0; b0 < a.B; b0 += BT) {
        const int bt = (a.B - b0) < BT ? (a.B - b0) : BT;
        // Software pipeline over K segments of KSEG: while segment s is multiplied, the activation chunks of s+1 are
        // already on their way to registers and, queued BEHIND them (vmcnt retires in order), the weight fragments
        // of s+1.  One group of U = KSEG/32/4 fragment loads per wave per segment.
        constexpr int U = KSEG / 32 / 4;
        constexpr int RPW = BT / 4;          // activation rows staged per wave
        constexpr int CPL = KSEG / 8 / 64;  // 16-byte chunks per lane per row
        uint4 wv[DEPTH][U], wv2[DEPTH][U], xv[RPW][CPL], gv[CPL];
        auto issue_x = [&](int k0, int ks) {
            const int nch = ks >> 3;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                gv[j] = make_uint4(0, 0, 0, 0);
                if (PRO == PRO_RMSNORM && (FULL || lane + 64 * j < nch)) gv[j] = reinterpret_cast<const uint4*>(a.nw)[(k0 >> 3) + lane + 64 * j];
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int bb = wave + 4 * r;
                if (FULL) {
                    const uint4* xr = x_row_ptr(a, b0 + (bb < bt ? bb : bt - 1)) + (k0 >> 3);
#pragma unroll
                    for (int j = 0; j < CPL; ++j) xv[r][j] = xr[lane + 64 * j];
                } else {
                    const uint4* xr = bb < bt ? x_row_ptr(a, b0 + bb) + (k0 >> 3) : nullptr;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        xv[r][j] = make_uint4(0, 0, 0, 0);
                        if (xr && lane + 64 * j < nch) xv[r][j] = xr[lane + 64 * j];
                    }
                }
            }
        };
        auto issue_w = [&](auto BUF, int k0, int ks) {
            constexpr int wb = decltype(BUF)::value;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = 32 * wave + 128 * u;
                if (FULL || kk < ks) {
                    wv[wb][u] = ldg_nt(reinterpret_cast<const uint4*>(wrow + k0 + kk + fk));
                    if (EPI == EPI_SILU_MUL) wv2[wb][u] = ldg_nt(reinterpret_cast<const uint4*>(wrow2 + k0 + kk + fk));
                }
            }
        };
        const int ks_first = a.K < KSEG ? a.K : KSEG;
        // one segment holds the whole row (K <= KSEG: the depth transformers, CosyVoice2): the row statistics are computed from the very
        // registers that are staged afterwards (same rows per wave, same chunk -> lane assignment and order as the loop below), with the
        // weight fragments requested right behind them — no second read of x, and the weights travel during the statistics
        const bool one_seg = PRO == PRO_RMSNORM && a.K <= KSEG && !a.no_one_seg;
        if (one_seg) {
            __syncthreads();
            issue_x(0, ks_first);
            issue_w(vox_ic<0>{}, 0, ks_first);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int b = wave + 4 * r;
                float ss = 0.0f;
#pragma unroll
                for (int j = 0; j < CPL; ++j)
                    if (FULL || lane + 64 * j < (ks_first >> 3)) ss = sq8(xv[r][j], ss);
                ss = butterfly<64>(ss);
                if (lane == 0 && b < bt) rinv[b] = 1.0f / sqrtf(ss / (float)a.K + a.eps);
            }
            __syncthreads();
        } else if (PRO == PRO_RMSNORM) {
            __syncthreads();
            {   // the wave's rows together: their chunk loads are in flight at once (a row's sum keeps its order: the lane's chunks
                // in increasing c, then the butterfly)
                constexpr int NR = BT / 4;
                const uint4* xr[NR];
                float ss[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int b = wave + 4 * r;
                    xr[r] = x_row_ptr(a, b0 + (b < bt ? b : bt - 1));
                    ss[r] = 0.0f;
                }
                for (int c = lane; c < (a.K >> 3); c += 64) {
                    uint4 v[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) v[r] = xr[r][c];
#pragma unroll
                    for (int r = 0; r < NR; ++r) ss[r] = sq8(v[r], ss[r]);
                }
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int b = wave + 4 * r;
                    const float t = butterfly<64>(ss[r]);
                    if (lane == 0 && b < bt) rinv[b] = 1.0f / sqrtf(t / (float)a.K + a.eps);
                }
            }
            __syncthreads();
        }
        f32x4_t acc[MT], acc2[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            acc2[m] = acc[m];
        }
        {
            const int ks0 = a.K < KSEG ? a.K : KSEG;
            if (!one_seg) { issue_x(0, ks0); issue_w(vox_ic<0>{}, 0, ks0); }
            if constexpr (DEPTH > 1) { if (KSEG < a.K) issue_w(vox_ic<1>{}, KSEG, (a.K - KSEG) < KSEG ? (a.K - KSEG) : KSEG); }
            if constexpr (DEPTH > 2) { if (2 * KSEG < a.K) issue_w(vox_ic<2>{}, 2 * KSEG, (a.K - 2 * KSEG) < KSEG ? (a.K - 2 * KSEG) : KSEG); }
        }
        auto body = [&](auto BUF, int k0) {
            constexpr int wb = decltype(BUF)::value;
            const int ks = (a.K - k0) < KSEG ? (a.K - k0) : KSEG;
            const int nch = ks >> 3;
            __syncthreads();   // the previous segment's MFMAs are done with the LDS tile
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int bb = wave + 4 * r;
                const float ri = PRO == PRO_RMSNORM ? rinv[FULL ? (bb < bt ? bb : bt - 1) : (bb < bt ? bb : 0)] : 0.0f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const int c = lane + 64 * j;
                    if (FULL || c < nch) {
                        uint4 o = xv[r][j];
                        if (PRO == PRO_RMSNORM && (FULL || bb < bt)) {
                            o = norm_chunk(o, gv[j], ri);
                            if (a.x_out && blockIdx.x == 0 && bb < bt)
                                reinterpret_cast<uint4*>(a.x_out + (size_t)(b0 + bb) * a.x_out_stride)[(k0 >> 3) + c] = o;
                        }
                        *reinterpret_cast<uint4*>(xs + (size_t)bb * LDK + c * 8) = o;
                    }
                }
            }
            __syncthreads();
            uint4 cw[U], cw2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cw[u] = wv[wb][u];
                if (EPI == EPI_SILU_MUL) cw2[u] = wv2[wb][u];
            }
            if (k0 + KSEG < a.K) {
                const int nks = (a.K - k0 - KSEG) < KSEG ? (a.K - k0 - KSEG) : KSEG;
                issue_x(k0 + KSEG, nks);
            }
            if (k0 + DEPTH * KSEG < a.K) {      // (behind the activations of the next segment: they are needed first)
                const int kd = k0 + DEPTH * KSEG, nkd = (a.K - kd) < KSEG ? (a.K - kd) : KSEG;
                issue_w(BUF, kd, nkd);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = 32 * wave + 128 * u;
                if (FULL || kk < ks) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const uint4 xa = *reinterpret_cast<const uint4*>(xs + (size_t)(m * 16 + fr) * LDK + kk + fk);
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(xa), as_bf8(cw[u]), acc[m], 0, 0, 0);
                        if (EPI == EPI_SILU_MUL)
                            acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(xa), as_bf8(cw2[u]), acc2[m], 0, 0, 0);
                    }
                }
            }
        };
        for (int k0 = 0; k0 < a.K;) {
            body(vox_ic<0>{}, k0); k0 += KSEG;
            if constexpr (DEPTH > 1) { if (k0 >= a.K) break; body(vox_ic<1>{}, k0); k0 += KSEG; }
            if constexpr (DEPTH > 2) { if (k0 >= a.K) break; body(vox_ic<2>{}, k0); k0 += KSEG; }
        }
        // cross-wave reduction (fixed order: wave 0 + 1 + 2 + 3), then wave 0 runs the epilogue
        __syncthreads();
        f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);   // the x tile is dead now: [3 waves][2 sets][MT][64 lanes]
        if (wave > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                red[(((wave - 1) * 2 + 0) * MT + m) * 64 + lane] = acc[m];
                if (EPI == EPI_SILU_MUL) red[(((wave - 1) * 2 + 1) * MT + m) * 64 + lane] = acc2[m];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[m] += red[((w * 2 + 0) * MT + m) * 64 + lane];
                    if (EPI == EPI_SILU_MUL) acc2[m] += red[((w * 2 + 1) * MT + m) * 64 + lane];
                }
        }
        // D: column n = lane&15, rows (lane>>4)*4 + r
        const int n = n0 + fr;
        if (wave == 0 && n < a.N) {
            const float bv = a.bias ? bf2f(a.bias[n]) : 0.0f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = m * 16 + (lane >> 4) * 4 + r;
                    if (b >= bt) continue;
                    const size_t oi = (size_t)(b0 + b) * a.N + n;
                    bf16_t o;
                    if (EPI == EPI_SILU_MUL) {
                        const float g = bfround(acc[m][r]), u = bfround(acc2[m][r]);
                        o = f2bf(bfround(silu_c(g)) * u);
                    } else {
                        float v = acc[m][r];
                        if (a.bias) v = v + bv;
                        o = f2bf(v);
                        if (EPI == EPI_SILU) o = f2bf(silu_c(bf2f(o)));
                        if (a.residual) o = f2bf(bf2f(a.residual[oi]) + bf2f(o));
                    }
                    a.y[oi] = o;
                }
        }
    }
}

template <int MT, int PRO, int EPI, bool FULL, int KSEG, int DEPTH = 1>
static int launch_linear_mfma_t(hipStream_t st, const LinArgs& a) {
    constexpr int BT = 16 * MT;
    const size_t smem = (size_t)BT * (KSEG + 8) * 2 + BT * 4;
    auto kern = k_linear_mfma<MT, PRO, EPI, FULL, KSEG, DEPTH>;
    if (smem > 64 * 1024) {
        static bool done = false;
        if (!done) {
            VOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((a.N + 15) / 16), dim3(256), smem, st, a);
    return VOX_OK;
}

// ================================================================================================
// k_linear_mfma without the LDS tile, for calls of at most 16 rows (one 16-row MFMA tile per block).  With a single row tile the four
// waves of a block multiply DISJOINT k-steps (wave w owns k = 32 w + 128 t .. + 31, t = 0, 1, ...: MF_KS = 8 x 128, so the 1024-wide
// segments of the staged kernel collapse into this one sequence), i.e. no activation element is shared between the waves and the LDS
// tile only adds two barriers and a write + read per segment.  Here a lane reads its A fragment (row lane & 15 of x, 16 bytes,
// L2-resident) the way it reads its B fragment, normalises it in registers when the call has a norm prologue, and the waves never
// meet before the sum of the four partial accumulators.  Each wave chains the same k-steps in the same order and the partials are
// added in wave order: bit-identical to k_linear_mfma (oracle: VR_ORD_MFMA4), which keeps the calls this form does not cover
// (17+ rows, x_out, a norm prologue with K > 1024).
//   k_linear_mfma_small   K <= 1024: every operand of the call is requested before anything waits (CosyVoice2's 896-wide linears)
//   k_linear_mfma_stream  K  > 1024, copy prologue: two register buffers of H k-steps per wave; a buffer is refilled as soon as it has
//                         been multiplied, the loop body has no conditional load (the in-order vmcnt accounting stays exact), steps
//                         past the end re-request the last real step (cache hits) and are not multiplied.  CosyVoice2's 4864-wide down
//                         projection has 56 column tiles: all of a block's 155 KB are in flight after two groups; GLM-4-Voice's
//                         13 696-wide one keeps 128 KB per CU in flight with no barrier in the walk.
// ================================================================================================
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void k_linear_mfma_small(LinArgs a) {
    constexpr bool SM = (EPI == EPI_SILU_MUL);
    __shared__ float rinv_s[16];
    __shared__ f32x4_t red[3][SM ? 2 : 1][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16, fr = lane & 15, fk = (lane >> 4) * 8;
    int nrow = n0 + fr;
    nrow = nrow < a.N ? nrow : a.N - 1;
    const int bt = a.B;                                     // <= 16
    const int T = a.K > 32 * wave ? (a.K - 32 * wave + 127) >> 7 : 0;      // this wave's k-steps (<= 8)
    // row statistics first (needed first, and loads retire in order): rows wave, wave + 4, ... — the lane's chunks in increasing c,
    // then the butterfly, as in k_linear_mfma
    uint4 sv[4][2];
    if (PRO == PRO_RMSNORM) {
        const int nch = a.K >> 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = wave + 4 * r;
            const uint4* xr = x_row_ptr(a, b < bt ? b : bt - 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                sv[r][j] = make_uint4(0, 0, 0, 0);
                if (lane + 64 * j < nch) sv[r][j] = xr[lane + 64 * j];
            }
        }
    }
    const uint4* xrow = x_row_ptr(a, fr < bt ? fr : bt - 1) + ((32 * wave + fk) >> 3);
    const uint4* gvp = reinterpret_cast<const uint4*>(a.nw) + ((32 * wave + fk) >> 3);
    const uint4* w0 = reinterpret_cast<const uint4*>(a.W + (size_t)nrow * a.K + 32 * wave + fk);
    const uint4* w1 = SM ? reinterpret_cast<const uint4*>(a.W2 + (size_t)nrow * a.K + 32 * wave + fk) : nullptr;
    uint4 xa[8], gv[8], wv[8], wv2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < T) {
            xa[t] = xrow[16 * t];
            if (PRO == PRO_RMSNORM) gv[t] = gvp[16 * t];
        }
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < T) {
            wv[t] = ldg_nt(w0 + 16 * t);
            if (SM) wv2[t] = ldg_nt(w1 + 16 * t);
        }
    float ri = 0.0f;
    if (PRO == PRO_RMSNORM) {
        const int nch = a.K >> 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = wave + 4 * r;
            float ss = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (lane + 64 * j < nch) ss = sq8(sv[r][j], ss);
            ss = butterfly<64>(ss);
            if (lane == 0 && b < bt) rinv_s[b] = 1.0f / sqrtf(ss / (float)a.K + a.eps);
        }
        __syncthreads();
        ri = rinv_s[fr < bt ? fr : bt - 1];
    }
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < T) {
            uint4 ax = xa[t];
            if (PRO == PRO_RMSNORM) ax = norm_chunk(ax, gv[t], ri);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(ax), as_bf8(wv[t]), acc, 0, 0, 0);
            if (SM) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(ax), as_bf8(wv2[t]), acc2, 0, 0, 0);
        }
    // residual of the outputs wave 0 finishes, requested before the waves meet
    float res_pre[4] = {0.f, 0.f, 0.f, 0.f};
    const int n = n0 + fr;
    if (!SM && a.residual && wave == 0 && n < a.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (lane >> 4) * 4 + r;
            if (b < bt) res_pre[r] = bf2f(a.residual[(size_t)b * a.N + n]);
        }
    }
    if (wave > 0) {
        red[wave - 1][0][lane] = acc;
        if (SM) red[wave - 1][SM ? 1 : 0][lane] = acc2;
    }
    __syncthreads();
    if (wave != 0 || n >= a.N) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        acc += red[w][0][lane];
        if (SM) acc2 += red[w][SM ? 1 : 0][lane];
    }
    const float bv = a.bias ? bf2f(a.bias[n]) : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = (lane >> 4) * 4 + r;       // D: column n = lane & 15, rows (lane >> 4) * 4 + r
        if (b >= bt) continue;
        const size_t oi = (size_t)b * a.N + n;
        bf16_t o;
        if (SM) {
            const float g = bfround(acc[r]), u = bfround(acc2[r]);
            o = f2bf(bfround(silu_c(g)) * u);
        } else {
            float v = acc[r];
            if (a.bias) v = v + bv;
            o = f2bf(v);
            if (EPI == EPI_SILU) o = f2bf(silu_c(bf2f(o)));
            if (a.residual) o = f2bf(res_pre[r] + bf2f(o));
        }
        a.y[oi] = o;
    }
}

template <int EPI, int H>
__global__ __launch_bounds__(256) void k_linear_mfma_stream(LinArgs a) {
    static_assert(EPI == EPI_STORE || EPI == EPI_SILU, "copy prologue, plain epilogues");
    __shared__ f32x4_t red[3][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16, fr = lane & 15, fk = (lane >> 4) * 8;
    int nrow = n0 + fr;
    nrow = nrow < a.N ? nrow : a.N - 1;
    const int bt = a.B;                                     // <= 16
    const int T = (a.K - 32 * wave + 127) >> 7;             // this wave's k-steps (K > 1024: at least 8)
    const int G = (T + H - 1) / H;
    const uint4* xrow = x_row_ptr(a, fr < bt ? fr : bt - 1) + ((32 * wave + fk) >> 3);     // step t: 16 chunks further
    // row-major: lane (fr, g) reads 16 B of weight row n0 + fr per step (16 rows x 64 B per wave request), step t 16 chunks further;
    // fragment-major (the decoder stack's copy, N % 16 == 0): the same register contents from ONE contiguous 1 KiB request — k-step
    // wave + 4 t of column tile blockIdx.x, four fragments further per step
    const bool wf = a.W_frag != nullptr;
    const int wstep = wf ? 256 : 16;
    const uint4* wrow = wf ? reinterpret_cast<const uint4*>(a.W_frag) + ((size_t)blockIdx.x * (a.K >> 5) + wave) * 64 + lane
                           : reinterpret_cast<const uint4*>(a.W + (size_t)nrow * a.K + 32 * wave + fk);
    uint4 xb[2][H], wb[2][H];
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto issue = [&](auto BUF, int g) {
        constexpr int b = decltype(BUF)::value;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            int t = g * H + i;
            t = t < T ? t : T - 1;                  // past the end: the last real step again (a cache hit; never multiplied)
            xb[b][i] = xrow[16 * t];
            wb[b][i] = ldg_nt(wrow + (size_t)wstep * t);
        }
    };
    auto consume = [&](auto BUF, auto GUARD, int g) {      // GUARD: the group may reach past the wave's last step
        constexpr int b = decltype(BUF)::value;
#pragma unroll
        for (int i = 0; i < H; ++i)
            if (!decltype(GUARD)::value || g * H + i < T)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(xb[b][i]), as_bf8(wb[b][i]), acc, 0, 0, 0);
    };
    issue(vox_ic<0>{}, 0);
    issue(vox_ic<1>{}, G > 1 ? 1 : 0);
    int g = 0;
    for (; g + 4 <= G; g += 2) {                    // steady state: every group multiplied here is complete, every one requested exists
        consume(vox_ic<0>{}, vox_ic<0>{}, g);
        issue(vox_ic<0>{}, g + 2);
        consume(vox_ic<1>{}, vox_ic<0>{}, g + 1);
        issue(vox_ic<1>{}, g + 3);
    }
    consume(vox_ic<0>{}, vox_ic<1>{}, g);           // one to three groups left; buffer 1 holds g + 1 (if it exists)
    issue(vox_ic<0>{}, g + 2 < G ? g + 2 : G - 1);
    consume(vox_ic<1>{}, vox_ic<1>{}, g + 1);
    consume(vox_ic<0>{}, vox_ic<1>{}, g + 2);
    float res_pre[4] = {0.f, 0.f, 0.f, 0.f};
    const int n = n0 + fr;
    if (a.residual && wave == 0 && n < a.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (lane >> 4) * 4 + r;
            if (b < bt) res_pre[r] = bf2f(a.residual[(size_t)b * a.N + n]);
        }
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave != 0 || n >= a.N) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) acc += red[w][lane];
    const float bv = a.bias ? bf2f(a.bias[n]) : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = (lane >> 4) * 4 + r;
        if (b >= bt) continue;
        const size_t oi = (size_t)b * a.N + n;
        float v = acc[r];
        if (a.bias) v = v + bv;
        bf16_t o = f2bf(v);
        if (EPI == EPI_SILU) o = f2bf(silu_c(bf2f(o)));
        if (a.residual) o = f2bf(res_pre[r] + bf2f(o));
        a.y[oi] = o;
    }
}
// VOX_MFMA_DIRECT=0: the staged kernel for every call (A/B timing; bit-identical)
static bool mfma_direct() {
    static const bool on = [] { const char* e = getenv("VOX_MFMA_DIRECT"); return !(e && e[0] == '0'); }();
    return on;
}

// K segment per LDS stage: 2048 from K = 4096 on (one 16-row tile: twice the weight bytes in flight per block and half the barriers of
// the 1024-wide walk; which k-steps a wave multiplies, and in which order, does not depend on the segment width, so the results are
// bit-identical).  VOX_MFMA_KSEG=1024 keeps the narrow walk (A/B timing).
static bool mfma_wide_seg() {
    static const bool on = [] { const char* e = getenv("VOX_MFMA_KSEG"); return !(e && atoi(e) == 1024); }();
    return on;
}
template <int PRO, int EPI>
static int launch_linear_mfma_pe(hipStream_t st, const LinArgs& a_in) {
    static const bool oneseg_off = [] { const char* e = getenv("VOX_MFMA_ONESEG"); return e && e[0] == '0'; }();
    LinArgs a = a_in;
    a.no_one_seg = oneseg_off ? 1 : 0;
    if (a.B <= 16 && !a.x_out && mfma_direct()) {      // one row tile: no LDS tile (same k-step chains, same wave-order sum)
        const dim3 grid((a.N + 15) / 16);
        if (a.K <= 1024) {
            hipLaunchKernelGGL((k_linear_mfma_small<PRO, EPI>), grid, dim3(256), 0, st, a);
            return VOX_OK;
        }
        if constexpr (PRO == PRO_COPY && (EPI == EPI_STORE || EPI == EPI_SILU)) {
            hipLaunchKernelGGL((k_linear_mfma_stream<EPI, 16>), grid, dim3(256), 0, st, a);
            return VOX_OK;
        }
    }
    // long K, plain store (GLM's down projection): three 1024-wide weight segments in flight — measured SLOWER than the 2048-wide walk
    // with one segment in flight (GLM-4-Voice B=8 LM step 5.60 -> 5.71 ms: the per-segment barriers and LDS writes, not the exposed
    // latency, are what the staged kernel pays); kept behind VOX_MFMA_DEPTH=3 as an A/B form, bit-identical
    static const bool deep = [] { const char* e = getenv("VOX_MFMA_DEPTH"); return e && atoi(e) == 3; }();
    if constexpr (PRO == PRO_COPY && EPI == EPI_STORE) {
        if (a.B <= 16 && a.K >= 4096 && deep) {
            const bool full1 = (a.K % MF_KS == 0) && (a.N % 16 == 0);
            return full1 ? launch_linear_mfma_t<1, PRO, EPI, true, MF_KS, 3>(st, a) : launch_linear_mfma_t<1, PRO, EPI, false, MF_KS, 3>(st, a);
        }
    }
    if (a.B <= 16 && a.K >= 4096 && mfma_wide_seg()) {
        const bool full2 = (a.K % 2048 == 0) && (a.N % 16 == 0);
        return full2 ? launch_linear_mfma_t<1, PRO, EPI, true, 2048>(st, a) : launch_linear_mfma_t<1, PRO, EPI, false, 2048>(st, a);
    }
    const bool full = (a.K % MF_KS == 0) && (a.N % 16 == 0);
    if (a.B <= 16) return full ? launch_linear_mfma_t<1, PRO, EPI, true, MF_KS>(st, a) : launch_linear_mfma_t<1, PRO, EPI, false, MF_KS>(st, a);
    return full ? launch_linear_mfma_t<2, PRO, EPI, true, MF_KS>(st, a) : launch_linear_mfma_t<2, PRO, EPI, false, MF_KS>(st, a);
}

__global__ void k_rmsnorm(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int H, float eps);
__global__ void k_rmsnorm_frag(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int H, float eps);

// ================================================================================================
// linear for 17+ rows where the full-K kernel below does not apply (33+ rows: prefill, depth step 1 of > 16 requests; or an
// unsupported K): split-K bf16 MFMA GEMM, LDS-tiled, weights read exactly once.
// (Fusing the slab reduction into the last-arriving block was tried: the device-scope fence it needs writes back and
// invalidates the XCD's L2 in every block — 3x slower than the separate k_splitk_reduce launch.)
// Grid (Ntot/64, K/512): a block owns 64 output columns and one 512-wide K slab for ALL rows of the pass (<= 128), walks
// the slab in 64-wide chunks through a double-buffered LDS tile pair (128-byte coalesced row segments of both the
// weights and the activations), wave w multiplying n-tile w against every row tile.  fp32 partial sums go to a
// workspace [slab][row][col]; k_splitk_reduce adds the slabs in slab order and runs the epilogue.  The K split is what
// keeps >= 128 blocks streaming at N = 2048 while each block still re-uses its activation tile for 64 columns.
template <int MT, int KS, bool SM>
__global__ __launch_bounds__(256) void k_gemm_splitk(LinArgs a, float* part, int b0, int bt, int rows_stride) {
    // one shot per block: the whole [64 x KS] weight slab and [BM x KS] activation slab are requested at once (every
    // thread has all its 16-byte loads in flight together: one exposed memory latency per block instead of one per
    // K chunk), parked in LDS, then multiplied.
    constexpr int BM = 16 * MT, BN = 64, LDK = KS + 8, SEG = KS / 8;
    constexpr int WL = BN * SEG / 256, XL = (BM * SEG + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ws = reinterpret_cast<bf16_t*>(smem);            // [BN][LDK]
    bf16_t* Xs = Ws + BN * LDK;                              // [BM][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const int Ntot = SM ? 2 * a.N : a.N;
    const int col0 = blockIdx.x * BN;
    const int k_begin = blockIdx.y * KS;
    const int klen = (a.K - k_begin) < KS ? (a.K - k_begin) : KS;      // multiple of 32
    const int nseg = klen >> 3;
    uint4 wreg[WL], xreg[XL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int idx = tid + 256 * i, r = idx / SEG, seg = idx % SEG;
        int col = col0 + r;
        col = col < Ntot ? col : Ntot - 1;
        const bf16_t* base = (SM && col >= a.N) ? a.W2 + (size_t)(col - a.N) * a.K : a.W + (size_t)col * a.K;
        wreg[i] = make_uint4(0, 0, 0, 0);
        if (seg < nseg) wreg[i] = ldg_nt(reinterpret_cast<const uint4*>(base + k_begin) + seg);
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int idx = tid + 256 * i, r = idx / SEG, seg = idx % SEG;
        xreg[i] = make_uint4(0, 0, 0, 0);
        if (idx < BM * SEG && seg < nseg)
            xreg[i] = (x_row_ptr(a, b0 + (r < bt ? r : bt - 1)) + (k_begin >> 3))[seg];
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int idx = tid + 256 * i;
        *reinterpret_cast<uint4*>(&Ws[(idx / SEG) * LDK + (idx % SEG) * 8]) = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int idx = tid + 256 * i;
        if (idx < BM * SEG) *reinterpret_cast<uint4*>(&Xs[(idx / SEG) * LDK + (idx % SEG) * 8]) = xreg[i];
    }
    __syncthreads();
    f32x4_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < klen; kk += 32) {
        const uint4 bw = *reinterpret_cast<const uint4*>(&Ws[(wave * 16 + fr) * LDK + kk + fk]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const uint4 ax = *reinterpret_cast<const uint4*>(&Xs[(m * 16 + fr) * LDK + kk + fk]);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(ax), as_bf8(bw), acc[m], 0, 0, 0);
        }
    }
    const int col = col0 + wave * 16 + fr;
    if (col < Ntot) {
        float* dst = part + (size_t)blockIdx.y * rows_stride * Ntot + col;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + (lane >> 4) * 4 + r;
                if (row < bt) dst[(size_t)row * Ntot] = acc[m][r];
            }
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* part, int S, int rows_stride, LinArgs a, int b0, int bt) {
    constexpr bool SM = (EPI == EPI_SILU_MUL);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= bt * a.N) return;
    const int row = idx / a.N, n = idx % a.N;
    const int Ntot = SM ? 2 * a.N : a.N;
    const float* p0 = part + (size_t)row * Ntot + n;
    float v = 0.0f, u = 0.0f;
    for (int sl = 0; sl < S; ++sl) {
        v = v + p0[(size_t)sl * rows_stride * Ntot];
        if (SM) u = u + p0[(size_t)sl * rows_stride * Ntot + a.N];
    }
    const size_t oi = (size_t)(b0 + row) * a.N + n;
    bf16_t o;
    if (SM) {
        o = f2bf(bfround(silu_c(bfround(v))) * bfround(u));
    } else {
        if (a.bias) v = v + bf2f(a.bias[n]);
        o = f2bf(v);
        if (EPI == EPI_SILU) o = f2bf(silu_c(bf2f(o)));
        if (a.residual) o = f2bf(bf2f(a.residual[oi]) + bf2f(o));
    }
    a.y[oi] = o;
}

// Row-wise variant: one block per output row adds the slabs, runs the epilogue AND applies the RMSNorm the next linear
// needs (post_out = norm(y) * post_nw): the separate k_rmsnorm launch of that linear disappears.  (Reduction tree of
// the sum of squares is the block's, not the canonical lane order: this path is the bf16-rounding-parity one.)
__global__ __launch_bounds__(1024) void k_splitk_reduce_rows(const float* part, int S, int rows_stride, LinArgs a, int b0) {
    __shared__ float red[16];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* p0 = part + (size_t)row * a.N;
    float ss = 0.0f;
    for (int n = tid; n < a.N; n += 1024) {
        float v = 0.0f;
        for (int sl = 0; sl < S; ++sl) v = v + p0[(size_t)sl * rows_stride * a.N + n];
        const size_t oi = (size_t)(b0 + row) * a.N + n;
        if (a.bias) v = v + bf2f(a.bias[n]);
        bf16_t o = f2bf(v);
        if (a.residual) o = f2bf(bf2f(a.residual[oi]) + bf2f(o));
        a.y[oi] = o;
        const float f = bf2f(o);
        ss = fmaf(f, f, ss);
    }
    ss = butterfly<64>(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
    for (int w = 0; w < 16; ++w) tot = tot + red[w];
    const float rinv = 1.0f / sqrtf(tot / (float)a.N + a.post_eps);
    for (int n = tid; n < a.N; n += 1024) {      // each thread re-reads exactly the y values it wrote
        const size_t oi = (size_t)(b0 + row) * a.N + n;
        a.post_out[oi] = f2bf((bf2f(a.y[oi]) * rinv) * bf2f(a.post_nw[n]));
    }
}

template <int MT, int KS, bool SM>
static void launch_gemm_splitk_t(hipStream_t st, const LinArgs& a, float* ws, int b0, int bt, int rs, dim3 grid) {
    const size_t smem = (size_t)(16 * MT + 64) * (KS + 8) * 2;
    auto kern = k_gemm_splitk<MT, KS, SM>;
    static bool done = false;
    if (!done && smem > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a, ws, b0, bt, rs);
}
template <int EPI>
static int launch_gemm_splitk(hipStream_t st, const LinArgs& a, float* ws, size_t ws_bytes) {
    constexpr bool SM = (EPI == EPI_SILU_MUL);
    const int Ntot = SM ? 2 * a.N : a.N;
    for (int b0 = 0; b0 < a.B; b0 += 128) {
        const int bt = (a.B - b0) < 128 ? (a.B - b0) : 128;
        const int mt = bt <= 16 ? 1 : bt <= 32 ? 2 : bt <= 64 ? 4 : bt <= 80 ? 5 : 8;
        const int ks = 256;          // slab width: two blocks (tiles of (16 MT + 64) x 264 bf16) share a CU's 160 KB LDS
        const int S = (a.K + ks - 1) / ks, rs = 16 * mt;
        if ((size_t)S * rs * Ntot * 4 > ws_bytes) return vox_fail(VOX_ERR_INVALID, "linear: split-K workspace too small");
        const dim3 grid((Ntot + 63) / 64, S);
        if (mt == 1) launch_gemm_splitk_t<1, 256, SM>(st, a, ws, b0, bt, rs, grid);
        else if (mt == 2) launch_gemm_splitk_t<2, 256, SM>(st, a, ws, b0, bt, rs, grid);
        else if (mt == 4) launch_gemm_splitk_t<4, 256, SM>(st, a, ws, b0, bt, rs, grid);
        else if (mt == 5) launch_gemm_splitk_t<5, 256, SM>(st, a, ws, b0, bt, rs, grid);
        else launch_gemm_splitk_t<8, 256, SM>(st, a, ws, b0, bt, rs, grid);
        if (EPI == EPI_STORE && a.post_nw && a.post_out)
            hipLaunchKernelGGL(k_splitk_reduce_rows, dim3(bt), dim3(1024), 0, st, ws, S, rs, a, b0);
        else
            hipLaunchKernelGGL((k_splitk_reduce<EPI>), dim3((bt * a.N + 255) / 256), dim3(256), 0, st, ws, S, rs, a, b0, bt);
    }
    return VOX_OK;
}

// ================================================================================================
// linear for 9..32 rows (batched decode): full-K bf16 MFMA GEMM, ONE launch per linear (no partial-sum workspace, no
// reduce kernel): at these row counts a frame is a chain of ~1000 dependent launches, so the second launch of the
// split-K pair costs as much as the GEMM itself.  A block owns 16 output columns (gate/up: 16 + 16) for all rows; its
// 8 waves split K into 8 contiguous ranges of KSTEPS*32, every lane requests ALL its weight fragments (one 16-byte
// load per k-step, straight into the B operand layout) before anything waits, activations follow in groups; partial
// accumulators meet in LDS and are added in wave order; prologue RMSNorm (sum of squares over the block's own
// operand registers) and the bias / residual / SiLU*up epilogue are fused.  bf16-rounding parity (MFMA order).
// ================================================================================================
// CT = 2 (gate / up of the 17..32-row talker: 384 column tiles, one 8-wave block per CU, i.e. a round and a half): a block owns two
// consecutive column tiles — 192 blocks, one round — and loads, squares and normalises its activation fragments ONCE for both; the
// second tile's weight fragments are requested when the first tile's MFMAs have been issued (into the same registers: both tiles'
// fragments at once do not fit the 256 registers a wave has at two waves per SIMD) and travel during the first tile's sum and
// epilogue.  Every output keeps its K split and summation order, so the results are bit-identical to CT = 1.
template <int MT, int KSTEPS, int PRO, int EPI, bool ROWSPLIT, int CT = 1>
__global__ __launch_bounds__(512) void k_gemm_fullk(LinArgs a) {
    static_assert(CT == 1 || (EPI == EPI_SILU_MUL && !ROWSPLIT), "two column tiles per block: the SiLU*up form only");
    constexpr bool SM = (EPI == EPI_SILU_MUL);
    constexpr int NB = SM ? 2 : 1;
    constexpr int G = (PRO == PRO_RMSNORM) ? KSTEPS : (KSTEPS <= 8 ? KSTEPS : 4);   // activation k-steps per group
    constexpr int NG = KSTEPS / G;
    __shared__ f32x4_t red[8][NB][MT][64];
    __shared__ float ssq[8][16 * MT];
    VOX_TR_DECL
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    // ROWSPLIT: row_tiles blocks per column tile, one per 16*MT-row tile, ids 8 apart so that they land on the same XCD and
    // the later readers of the weight tile hit that XCD's L2 (17..32 rows with few column tiles: 2 x 16 rows on twice the
    // CUs; 33..64 rows: 2 x 32 rows).
    int ctile = blockIdx.x * CT, r0 = 0;
    if (ROWSPLIT) {
        ctile = (blockIdx.x / (8 * a.row_tiles)) * 8 + (blockIdx.x & 7);
        r0 = ((blockIdx.x >> 3) % a.row_tiles) * (a.rt_rows ? a.rt_rows : 16 * MT);
    }
    const int n0 = ctile * 16;
    const int kbase = wave * (KSTEPS * 32) + fk;
    // Sub-tile row split (rt_rows = 8 | 4, MT = 1): a linear of <= 16 rows with few column tiles (N / 16 <= 128: 64 .. 128 blocks on 256
    // CUs) runs 2 .. 4 blocks per column tile, each owning rt_rows of the rows — its MFMA row tile is padded with copies of its last row,
    // whose outputs are dropped.  An output element's K split and summation order do not depend on which rows share its tile, so the
    // results are bit-identical; per block the activation traffic (as many bytes as the weight tile at 16 rows) shrinks with the rows.
    const int bt = (ROWSPLIT && a.rt_rows && a.B - r0 > a.rt_rows) ? a.rt_rows : a.B - r0;
    uint4 wv[NB][KSTEPS];
    const size_t fbase = (size_t)wave * KSTEPS * 64 + lane;      // uint4 offset of this wave's first fragment inside a tile row
    // row-major: lane (fr, g) reads 16 B of weight row n0+fr per k-step (16 rows x 64 B per wave request);
    // fragment-major: the same register contents from ONE contiguous 1 KiB request
    const bool wf = a.W_frag != nullptr;
    const int wstep = wf ? 64 : 4;
    const uint4* w0 = wf ? reinterpret_cast<const uint4*>(a.W_frag) + (size_t)ctile * (a.K >> 5) * 64 + fbase
                         : reinterpret_cast<const uint4*>(a.W + (size_t)((n0 + fr) < a.N ? (n0 + fr) : a.N - 1) * a.K + kbase);   // (N % 16 tail: clamped, not stored)
    const uint4* w1 = !SM ? nullptr
                      : wf ? reinterpret_cast<const uint4*>(a.W2_frag) + (size_t)ctile * (a.K >> 5) * 64 + fbase
                           : reinterpret_cast<const uint4*>(a.W2 + (size_t)(n0 + fr) * a.K + kbase);
    // the next column tile: (K / 32) 1 KiB fragments further in fragment-major form, 16 weight rows further row-major
    const size_t tstride = wf ? (size_t)(a.K >> 5) * 64 : (size_t)16 * (a.K >> 3);
    auto load_w = [&](int ct) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            wv[0][s] = a.keep ? w0[ct * tstride + s * wstep] : ldg_nt(w0 + ct * tstride + s * wstep);
            if (SM) wv[1][s] = a.keep ? w1[ct * tstride + s * wstep] : ldg_nt(w1 + ct * tstride + s * wstep);
        }
    };
    // Norm-prologue form: the activation fragments are requested BEFORE the weight tile (VOX_XFIRST, default on).  Loads return in order,
    // and the row statistics (the block's first barrier) need the activations only: asked for last, they arrived behind the whole weight
    // tile (tools/chain_trace.py: "x arrived" 2.1 us after "all loads issued" in the 32-row talker GEMMs); asked for first, the sum of
    // squares, the barrier and the normalisation run while the weights are still streaming.  Same loads, same arithmetic.
    constexpr bool XF = VOX_XFIRST >= 2 || ((PRO == PRO_RMSNORM) && VOX_XFIRST != 0);
    const uint4* xr[MT];
    const int xstep = a.x_frag ? 64 : 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + fr;
        const int arow = r0 + (row < bt ? row : bt - 1);       // (fragment-major: row arow sits in tile arow / 16 at fragment lane arow % 16 + 16 * (k group))
        xr[m] = a.x_frag ? reinterpret_cast<const uint4*>(a.x_frag) + (size_t)(arow >> 4) * (a.K >> 5) * 64 + (size_t)wave * KSTEPS * 64 + (arow & 15) + (lane & 48)
                         : x_row_ptr(a, arow) + (kbase >> 3);
    }
    static_assert(CT == 1 || (PRO == PRO_RMSNORM), "two column tiles: every activation fragment resident (norm prologue form)");
    f32x4_t acc[NB][MT];
    uint4 xa[NG > 1 ? 2 : 1][MT][G];
    uint4 gv[PRO == PRO_RMSNORM ? G : 1];
    auto load_x0 = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < G; ++s) xa[0][m][s] = xr[m][s * xstep];
        if (PRO == PRO_RMSNORM) {
            const uint4* nw = reinterpret_cast<const uint4*>(a.nw) + (kbase >> 3);
#pragma unroll
            for (int s = 0; s < G; ++s) gv[s] = nw[s * 4];
        }
    };
    if (XF) load_x0();
    load_w(0);
    // residual of the outputs this thread will finish (threads < MT*64), requested with the operands instead of after the reduce
    // (branch-free: rows past the tile's last are clamped and their values dropped — a conditional load is joined with its default by
    // a copy behind a wait of its own, i.e. the four requests became four dependent round trips in front of this wave's MFMAs)
    u32 res_raw[4], bias_raw;      // bf16 bits, zero-extended by the load itself (no defaults, nothing to compute on arrival: no wait here)
    if (!SM && __builtin_amdgcn_readfirstlane(wave) < MT) {       // (a scalar branch: nothing below is predicated)
        const int nc = n0 + fr < a.N ? n0 + fr : a.N - 1;
        if (a.residual) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = wave * 16 + (lane >> 4) * 4 + r;
                res_raw[r] = a.residual[(size_t)(r0 + (b < bt ? b : bt - 1)) * a.N + nc];
            }
        }
        if (CT == 1 && a.bias) bias_raw = a.bias[nc];
    }
    if (!XF) load_x0();
    __builtin_amdgcn_sched_barrier(0);      // every load above is issued before anything below waits on one of them
    VOX_TR(1)
    float rinv[MT];
    if (PRO == PRO_RMSNORM) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float ss = 0.0f;
#pragma unroll
            for (int s = 0; s < G; ++s) ss = sq8(xa[0][m][s], ss);
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (lane < 16) ssq[wave][m * 16 + fr] = ss;
        }
        __syncthreads();
        VOX_TR(2)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += ssq[w][m * 16 + fr];
            rinv[m] = 1.0f / sqrtf(t / (float)a.K + a.eps);
        }
        if (CT > 1) {      // two column tiles: normalise the fragments in place, once, and let the norm weights' registers go
#pragma unroll
            for (int s = 0; s < G; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) xa[0][m][s] = norm_chunk(xa[0][m][s], gv[s], rinv[m]);
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[nb][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int s = 0; s < G; ++s) xa[(g + 1) & 1][m][s] = xr[m][((g + 1) * G + s) * xstep];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < G; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    uint4 ax = xa[g & 1][m][s];
                    if (PRO == PRO_RMSNORM && CT == 1) ax = norm_chunk(ax, gv[s], rinv[m]);   // normalised right before use: no second copy
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(ax), as_bf8(wv[nb][g * G + s]), acc[nb][m], 0, 0, 0);
                }
        }
        if (ct + 1 < CT) {      // the next tile's weights travel during this tile's sum and epilogue
            __builtin_amdgcn_sched_barrier(0);
            load_w(ct + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ct > 0) __syncthreads();      // the sum buffer is reused by the block's next column tile
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int m = 0; m < MT; ++m) red[wave][nb][m][lane] = acc[nb][m];
        if (ct == 0) { VOX_TR(3) }
        __syncthreads();
        if (ct == 0) { VOX_TR(4) }
        if (tid < MT * 64) {
            const int m = tid >> 6;            // thread (m, lane) finishes D fragment m: column n0+fr, rows m*16 + (lane>>4)*4 + r
            f32x4_t v = red[0][0][m][lane], u = SM ? red[0][NB - 1][m][lane] : v;
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                v += red[w][0][m][lane];
                if (SM) u += red[w][NB - 1][m][lane];
            }
            const int n = n0 + 16 * ct + fr;
            if (n < a.N) {           // (tail columns of an N that is not a multiple of 16: vocabulary heads)
                const float bv = a.bias ? bf2f((bf16_t)(CT == 1 && !SM ? bias_raw : (u32)a.bias[n])) : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = m * 16 + (lane >> 4) * 4 + r;
                    if (b >= bt) continue;
                    const size_t oi = (size_t)(r0 + b) * a.N + n;
                    bf16_t o;
                    if (SM) {
                        o = f2bf(bfround(silu_c(bfround(v[r]))) * bfround(u[r]));
                    } else {
                        float f = v[r];
                        if (a.bias) f = f + bv;
                        o = f2bf(f);
                        if (EPI == EPI_SILU) o = f2bf(silu_c(bf2f(o)));
                        if (a.residual) o = f2bf(bf2f((bf16_t)res_raw[r]) + bf2f(o));
                    }
                    if (a.y_rowmajor) a.y[oi] = o;
                    if (a.y_frag) a.y_frag[frag_off(r0 + b, n, a.N)] = o;
                }
            }
        }
    }
    VOX_TR_END(1, (MT << 24) | (KSTEPS << 16) | (PRO << 12) | (EPI << 8) | (ROWSPLIT ? 16 : 0) | CT)
}

// row-major [rows][K] -> fragment-major (one thread per 16-byte chunk)
__global__ __launch_bounds__(256) void k_swizzle_frag(const uint4* src, uint4* dst, int rows, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;       // destination chunk index
    const size_t total = (size_t)rows * (K >> 3);
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const size_t frag = i >> 6;
    const int ks = (int)(frag % (K >> 5)), t = (int)(frag / (K >> 5));
    const int r = t * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8;
    dst[i] = src[((size_t)r * K + k) >> 3];
}
int vox_launch_swizzle_frag(hipStream_t st, const void* src, void* dst, int rows, int K) {
    if (rows % 16 || K % 32) return vox_fail(VOX_ERR_INVALID, "swizzle_frag: rows %% 16 or K %% 32");
    const size_t total = (size_t)rows * (K >> 3);
    hipLaunchKernelGGL(k_swizzle_frag, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, rows, K);
    return VOX_OK;
}

// Development probe (VOX_PRENORM2048=1, dev builds only): RMSNorm of the 17..32-row talker linears as its own launch + copy-prologue
// GEMM ("normalise once").  Measured on MI355X, B = 32: talker 1.95 -> 2.27 ms with a row-major normed operand, 2.08 ms fragment-major:
// the two extra launches per layer cost more than the per-block prologue they replace.  Not used by the shipped library.
#ifdef VOX_DEV_KNOBS
static bool dev_prenorm2048() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VOX_PRENORM2048"); v = e ? atoi(e) : 0; }
    return v != 0;
}
#else
static constexpr bool dev_prenorm2048() { return false; }
#endif
static bool fullk_shape_ok(int B, int N, int K, int pro, int epi, int exact_rows = 2) {
    if (B <= exact_rows || B < 2 || B > 128 || K % 256) return false;
    if (N % 16 && (B > 32 || epi == EPI_SILU_MUL)) return false;    // a column tail only for plain outputs of <= 32 rows (row-major weights)
    if (B > 32 && N % 128) return false;          // 33..128 rows: 2..4 32-row tiles per column tile (block ids 8 apart)
    const int ks = K / 256;
    if (pro == PRO_RMSNORM && (epi == EPI_STORE || epi == EPI_SILU_MUL)) return ks == 4 || ks == 8;
    if (pro == PRO_COPY && epi == EPI_STORE) return ks == 4 || ks == 8 || ks == 12 || ks == 16 || ks == 24 || ks == 32;
    if (pro == PRO_COPY && epi == EPI_SILU_MUL) return ks == 16 || (ks == 8 && dev_prenorm2048());
    return false;
}
// K = 4096 with a norm prologue (GLM-4-Voice width): the operand registers of a whole row slice do not fit next to the weights,
// so the row is normalised once into the caller's scratch and the copy-prologue kernel runs on that
static bool fullk_prenorm_ok(const LinearCall& c) {
    return !c.fixed_order && !c.x_out && !c.x_rows && c.norm_scratch && c.pro == PRO_RMSNORM && (c.K == 4096 || (c.K == 2048 && c.B > 16 && dev_prenorm2048())) &&
           (c.epi == EPI_STORE || c.epi == EPI_SILU_MUL) && fullk_shape_ok(c.B, c.N, c.K, PRO_COPY, c.epi == EPI_SILU_MUL ? EPI_SILU_MUL : EPI_STORE, c.exact_rows);
}
template <int MT, int KSTEPS, int PRO, int EPI>
static int launch_gemm_fullk_t(hipStream_t st, const LinArgs& a0) {
    LinArgs a = a0;
    a.row_tiles = 1;
    a.rt_rows = 0;
    if constexpr (MT == 1 && PRO == PRO_COPY) {
        // <= 16 rows, few column tiles: 2 (4) blocks of 8 (4) rows per column tile (k_gemm_fullk: sub-tile row split); VOX_ROWSPLIT_SUB=0: off
        static const int sub = [] { const char* e = getenv("VOX_ROWSPLIT_SUB"); return e ? atoi(e) : 1; }();
        const int tiles = a.N / 16;
        if (sub && a.N % 16 == 0 && tiles % 8 == 0 && tiles <= 128 && a.B > 4) {
            a.rt_rows = (tiles <= 64 && a.B > 8 && sub != 8) ? 4 : (a.B > 8 ? 8 : 4);
            if (sub == 4 && a.B > 4) a.rt_rows = 4;
            a.row_tiles = (a.B + a.rt_rows - 1) / a.rt_rows;
            if (a.row_tiles > 1) {
                hipLaunchKernelGGL((k_gemm_fullk<1, KSTEPS, PRO, EPI, true>), dim3(tiles * a.row_tiles), dim3(512), 0, st, a);
                return VOX_OK;
            }
            a.row_tiles = 1; a.rt_rows = 0;
        }
    }
    if constexpr (MT == 2) {
        if (a.B > 32) {            // 33..128 rows (short prefills, depth step 1 of 17+ requests): 32-row blocks side by side
            a.row_tiles = (a.B + 31) / 32;
            hipLaunchKernelGGL((k_gemm_fullk<2, KSTEPS, PRO, EPI, true>), dim3(a.N / 16 * a.row_tiles), dim3(512), 0, st, a);
            return VOX_OK;
        }
        if constexpr (PRO == PRO_COPY) {
            // few column tiles: per-CU load rate is the limit, so use twice the CUs (two 16-row blocks per column tile)
            if (a.N % 16 == 0 && a.N / 16 <= 128 && (a.N / 16) % 8 == 0) {
                a.row_tiles = 2;
                // <= 64 column tiles (the depth loop's o_proj / down: N = 1024): two 16-row blocks per tile still leave half the CUs idle —
                // four blocks of 8 rows (the sub-tile split of the <= 16-row case: padded MFMA rows, outputs dropped, bit-identical) put
                // a block on every CU and halve each block's activation bytes (VOX_ROWSPLIT_SUB32=0: two 16-row blocks)
                static const bool sub32 = [] { const char* e = getenv("VOX_ROWSPLIT_SUB32"); return !(e && e[0] == '0'); }();
                if (sub32 && a.N / 16 <= 64 && a.B > 24) { a.rt_rows = 8; a.row_tiles = (a.B + 7) / 8; }
                hipLaunchKernelGGL((k_gemm_fullk<1, KSTEPS, PRO, EPI, true>), dim3(a.N / 16 * a.row_tiles), dim3(512), 0, st, a);
                return VOX_OK;
            }
        }
    }
    if constexpr (KSTEPS == 8 && PRO == PRO_RMSNORM && EPI == EPI_SILU_MUL) {
        // more column tiles than one round of 8-wave blocks (one per CU): two tiles per block (VOX_FULLK_CT2=0: one)
        static const bool ct2 = [] { const char* e = getenv("VOX_FULLK_CT2"); return !(e && e[0] == '0'); }();
        if (ct2 && a.N % 32 == 0 && a.N / 16 > 256 && a.N / 16 <= 512) {
            hipLaunchKernelGGL((k_gemm_fullk<MT, KSTEPS, PRO, EPI, false, 2>), dim3(a.N / 32), dim3(512), 0, st, a);
            return VOX_OK;
        }
    }
    hipLaunchKernelGGL((k_gemm_fullk<MT, KSTEPS, PRO, EPI, false>), dim3((a.N + 15) / 16), dim3(512), 0, st, a);
    return VOX_OK;
}
template <int PRO, int EPI>
static int launch_gemm_fullk(hipStream_t st, const LinArgs& a) {
    const int ks = a.K / 256;
#define VOX_FK(KS) if (ks == KS) return a.B <= 16 ? launch_gemm_fullk_t<1, KS, PRO, EPI>(st, a) : launch_gemm_fullk_t<2, KS, PRO, EPI>(st, a);
    if constexpr (!(PRO == PRO_COPY && EPI == EPI_SILU_MUL)) { VOX_FK(4) VOX_FK(8) }
    if constexpr (PRO == PRO_COPY && EPI == EPI_STORE) { VOX_FK(12) VOX_FK(16) VOX_FK(24) VOX_FK(32) }
    if constexpr (PRO == PRO_COPY && EPI == EPI_SILU_MUL) { VOX_FK(8) VOX_FK(16) }
#undef VOX_FK
    return vox_fail(VOX_ERR_INVALID, "linear(full-K): unsupported K");
}
// true when this call takes the one-launch full-K path (9..32 rows, K a supported multiple of 256)
static bool linear_is_fullk(const LinearCall& c) {
    return (!c.fixed_order && !c.x_out && fullk_shape_ok(c.B, c.N, c.K, c.pro, c.epi, c.exact_rows)) || fullk_prenorm_ok(c);
}
bool vox_linear_is_fullk(const LinearCall& c) { return linear_is_fullk(c); }
bool vox_fullk_weight_ok(int N, int K) {
    const int ks = K / 256;
    return N % 16 == 0 && K % 256 == 0 && (ks == 4 || ks == 8 || ks == 12 || ks == 16 || ks == 24 || ks == 32);
}
// ... and for k_linear_mfma_stream (<= 16 rows, copy prologue, K > 1024 off the full-K shapes: GLM-4-Voice's / CosyVoice2's down projection)
bool vox_stream_weight_ok(int N, int K) { return !vox_fullk_weight_ok(N, K) && N % 16 == 0 && K % 32 == 0 && K > 1024; }

static int rows_gemm_min() {
#ifdef VOX_DEV_KNOBS
    static int rows_min = -1;   // development builds: smallest row count routed to the split-K GEMM
    if (rows_min < 0) { const char* e = getenv("VOX_ROWS_MIN"); rows_min = e ? atoi(e) : 17; }
    return rows_min;
#else
    return 17;                  // 9..16 rows: one 16-row MFMA tile per block (k_linear_mfma) is faster (measured)
#endif
}
// true when this call takes the 17+ rows split-K path (the only one that honours post_norm_* / x_prenormed)
bool vox_linear_is_rows_gemm(const LinearCall& c) {
    return !linear_is_fullk(c) && c.B >= rows_gemm_min() && !c.fixed_order && c.K % 32 == 0 && c.splitk_ws != nullptr && !c.x_out &&
           (c.pro == PRO_COPY || (c.pro == PRO_RMSNORM && (c.x_prenormed || (c.norm_scratch && !c.x_rows && (c.x_stride == 0 || c.x_stride == c.K)))));
}

int vox_launch_linear(vox_ctx* ctx, hipStream_t st, const LinearCall& call) {
    LinearCall c = call;
    c.exact_rows = ctx->exact_rows;       // the context's setting is authoritative (callers that plan hand-offs copy it too)
    if (c.K % 8 != 0 || c.B <= 0 || c.N <= 0) return vox_fail(VOX_ERR_INVALID, "linear: K%8!=0 or empty");
    LinArgs a{};
    a.W = (const bf16_t*)c.W; a.W2 = (const bf16_t*)c.W2; a.bias = (const bf16_t*)c.bias;
    a.x = (const bf16_t*)c.x; a.residual = (const bf16_t*)c.residual; a.nw = (const bf16_t*)c.norm_w;
    a.y = (bf16_t*)c.y; a.x_out = (bf16_t*)c.x_out; a.part_o = c.part_o; a.part_ml = c.part_ml;
    a.kvlen = c.kvlen; a.x_rows = c.x_rows; a.x_stride = c.x_stride ? c.x_stride : c.K;
    a.x_out_stride = c.x_out_stride ? c.x_out_stride : c.K; a.eps = c.eps; a.B = c.B; a.N = c.N; a.K = c.K; a.Hq = c.Hq; a.D = c.D;
    a.max_chunks = c.max_chunks;
    a.keep = c.keep_weights;
    a.post_nw = (const bf16_t*)c.post_norm_w; a.post_out = (bf16_t*)c.post_norm_out; a.post_eps = c.eps;
    a.y_rowmajor = 1;
    if (linear_is_fullk(c)) {
        a.W_frag = (const bf16_t*)c.W_frag; a.W2_frag = (const bf16_t*)c.W2_frag; a.x_frag = (const bf16_t*)c.x_frag;
        a.y_frag = (bf16_t*)c.y_frag; a.y_rowmajor = c.y_rowmajor || !c.y_frag;
        if (c.epi == EPI_SILU_MUL && (a.W_frag == nullptr) != (a.W2_frag == nullptr))
            return vox_fail(VOX_ERR_INVALID, "linear: W_frag and W2_frag must be given together");
    } else if (c.x_frag && !c.x) {
        return vox_fail(VOX_ERR_INVALID, "linear: fragment-major input outside the 9..32 rows path");
    }
    const int ncu = ctx->n_cu;
    int pro = c.pro, epi = c.epi;
#ifdef VOX_DEV_KNOBS        // development builds: strip features to time them (results are wrong when set)
    static int dev = -1;
    if (dev < 0) { const char* e = getenv("VOX_DEV"); dev = e ? atoi(e) : 0; }
    if ((dev & 1) && pro == PRO_RMSNORM) pro = PRO_COPY;
    if (dev & 2) a.residual = nullptr;
    if ((dev & 4) && epi == EPI_SILU_MUL) epi = EPI_STORE;
    if (dev & 8) a.bias = nullptr;
#endif
    if (fullk_prenorm_ok(c) && pro == PRO_RMSNORM) {
#ifdef VOX_DEV_KNOBS
        if (c.K == 2048 && dev_prenorm2048()) {      // timing probe: the normalised rows written fragment-major
            static bf16_t* nfrag = nullptr;
            if (!nfrag) (void)hipMalloc((void**)&nfrag, (size_t)128 * 4096 * 2);
            hipLaunchKernelGGL(k_rmsnorm_frag, dim3((c.B + 3) / 4), dim3(256), 0, st, a.x, a.nw, nfrag, c.B, c.K, c.eps);
            a.x = (const bf16_t*)c.norm_scratch; a.x_stride = c.K; a.x_frag = nfrag;
            if (epi == EPI_SILU_MUL) return launch_gemm_fullk<PRO_COPY, EPI_SILU_MUL>(st, a);
            return launch_gemm_fullk<PRO_COPY, EPI_STORE>(st, a);
        }
#endif
        hipLaunchKernelGGL(k_rmsnorm, dim3((c.B + 3) / 4), dim3(256), 0, st, a.x, a.nw, (bf16_t*)c.norm_scratch, c.B, c.K, c.eps);
        a.x = (const bf16_t*)c.norm_scratch; a.x_stride = c.K; a.x_frag = nullptr;
        if (epi == EPI_SILU_MUL) return launch_gemm_fullk<PRO_COPY, EPI_SILU_MUL>(st, a);
        return launch_gemm_fullk<PRO_COPY, EPI_STORE>(st, a);
    }
    if (linear_is_fullk(c)) {
        if (pro == PRO_RMSNORM && epi == EPI_STORE) return launch_gemm_fullk<PRO_RMSNORM, EPI_STORE>(st, a);
        if (pro == PRO_RMSNORM && epi == EPI_SILU_MUL) return launch_gemm_fullk<PRO_RMSNORM, EPI_SILU_MUL>(st, a);
        if (pro == PRO_COPY && epi == EPI_STORE) return launch_gemm_fullk<PRO_COPY, EPI_STORE>(st, a);
    }
    if (vox_linear_is_rows_gemm(c) && (pro == PRO_COPY || pro == PRO_RMSNORM)) {
        // 17+ rows: normalise once (not in every block), then the split-K MFMA GEMM
        if (pro == PRO_RMSNORM && c.x_prenormed) {
            a.x = (const bf16_t*)c.x_prenormed;       // the producing linear already wrote norm(x) (post_norm_out)
            a.x_stride = c.K;
        } else if (pro == PRO_RMSNORM) {
            if (a.x_rows) return vox_fail(VOX_ERR_INVALID, "linear: row indirection with a norm prologue at > 32 rows");
            hipLaunchKernelGGL(k_rmsnorm, dim3((c.B + 3) / 4), dim3(256), 0, st, a.x, a.nw, (bf16_t*)c.norm_scratch, c.B, c.K, c.eps);
            if (a.x_out) return vox_fail(VOX_ERR_INVALID, "linear: x_out with a norm prologue at > 32 rows");
            a.x = (const bf16_t*)c.norm_scratch;
            a.x_stride = c.K;
        }
        if (epi == EPI_STORE) return launch_gemm_splitk<EPI_STORE>(st, a, (float*)c.splitk_ws, c.splitk_ws_bytes);
        if (epi == EPI_SILU) return launch_gemm_splitk<EPI_SILU>(st, a, (float*)c.splitk_ws, c.splitk_ws_bytes);
        if (epi == EPI_SILU_MUL) return launch_gemm_splitk<EPI_SILU_MUL>(st, a, (float*)c.splitk_ws, c.splitk_ws_bytes);
    }
    if (c.B > c.exact_rows && !c.fixed_order && c.pro != PRO_ATTN && c.K % 32 == 0) {   // above exact_rows: MFMA path (weights streamed once per 32-row tile)
        static const bool stream_frag = [] { const char* e = getenv("VOX_STREAM_FRAG"); return !(e && e[0] == '0'); }();     // (=0: row-major weights, A/B timing)
        if (stream_frag && c.W_frag && c.B <= 16 && c.pro == PRO_COPY && c.epi != EPI_SILU_MUL && vox_stream_weight_ok(c.N, c.K)) a.W_frag = (const bf16_t*)c.W_frag;
#define VOX_PM(P, E) if (c.pro == P && c.epi == E) return launch_linear_mfma_pe<P, E>(st, a);
        VOX_PM(PRO_COPY, EPI_STORE) VOX_PM(PRO_COPY, EPI_SILU) VOX_PM(PRO_COPY, EPI_SILU_MUL)
        VOX_PM(PRO_RMSNORM, EPI_STORE) VOX_PM(PRO_RMSNORM, EPI_SILU_MUL)
#undef VOX_PM
        return vox_fail(VOX_ERR_INVALID, "linear(mfma): unsupported prologue/epilogue combination");
    }
#define VOX_PE(P, E) if (pro == P && epi == E) return launch_linear_pe<P, E>(st, a, ncu);
    VOX_PE(PRO_COPY, EPI_STORE) VOX_PE(PRO_COPY, EPI_SILU) VOX_PE(PRO_COPY, EPI_SILU_MUL)
    VOX_PE(PRO_RMSNORM, EPI_STORE) VOX_PE(PRO_RMSNORM, EPI_SILU_MUL) VOX_PE(PRO_ATTN, EPI_STORE)
#undef VOX_PE
    return vox_fail(VOX_ERR_INVALID, "linear: unsupported prologue/epilogue combination");
}

// ================================================================================================
// standalone RMSNorm (one wave per row)
// ================================================================================================
__global__ __launch_bounds__(256) void k_rmsnorm(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int H, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = H >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x) + (size_t)row * nch;
    const uint4* nw = reinterpret_cast<const uint4*>(w);
    uint4* yr = reinterpret_cast<uint4*>(y) + (size_t)row * nch;
    float s = 0.0f;
    if (nch <= 512) {
        // rows of up to 4096 values: the lane's (at most 8) chunks and their weights are requested together and stay in registers for
        // the scaling pass — one trip to memory instead of a dependent chain of loads per pass (same order of the sum: the lane's chunks
        // in increasing c, then the butterfly)
        uint4 v[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = lane + 64 * u;
            v[u] = make_uint4(0, 0, 0, 0); g[u] = v[u];
            if (c < nch) { v[u] = xr[c]; g[u] = nw[c]; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (lane + 64 * u < nch) s = sq8(v[u], s);
        s = butterfly<64>(s);
        const float rinv = 1.0f / sqrtf(s / (float)H + eps);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = lane + 64 * u;
            if (c < nch) {
                uint4 o;
                o.x = (u32)f2bf((bflo(v[u].x) * rinv) * bflo(g[u].x)) | ((u32)f2bf((bfhi(v[u].x) * rinv) * bfhi(g[u].x)) << 16);
                o.y = (u32)f2bf((bflo(v[u].y) * rinv) * bflo(g[u].y)) | ((u32)f2bf((bfhi(v[u].y) * rinv) * bfhi(g[u].y)) << 16);
                o.z = (u32)f2bf((bflo(v[u].z) * rinv) * bflo(g[u].z)) | ((u32)f2bf((bfhi(v[u].z) * rinv) * bfhi(g[u].z)) << 16);
                o.w = (u32)f2bf((bflo(v[u].w) * rinv) * bflo(g[u].w)) | ((u32)f2bf((bfhi(v[u].w) * rinv) * bfhi(g[u].w)) << 16);
                yr[c] = o;
            }
        }
        return;
    }
    for (int c = lane; c < nch; c += 64) s = sq8(xr[c], s);
    s = butterfly<64>(s);
    const float rinv = 1.0f / sqrtf(s / (float)H + eps);
    for (int c = lane; c < nch; c += 64) {
        uint4 v = xr[c], g = nw[c], o;
        o.x = (u32)f2bf((bflo(v.x) * rinv) * bflo(g.x)) | ((u32)f2bf((bfhi(v.x) * rinv) * bfhi(g.x)) << 16);
        o.y = (u32)f2bf((bflo(v.y) * rinv) * bflo(g.y)) | ((u32)f2bf((bfhi(v.y) * rinv) * bfhi(g.y)) << 16);
        o.z = (u32)f2bf((bflo(v.z) * rinv) * bflo(g.z)) | ((u32)f2bf((bfhi(v.z) * rinv) * bfhi(g.z)) << 16);
        o.w = (u32)f2bf((bflo(v.w) * rinv) * bflo(g.w)) | ((u32)f2bf((bfhi(v.w) * rinv) * bfhi(g.w)) << 16);
        yr[c] = o;
    }
}

// (development probe) the same, output fragment-major
__global__ __launch_bounds__(256) void k_rmsnorm_frag(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int H, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = H >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x) + (size_t)row * nch;
    const uint4* nw = reinterpret_cast<const uint4*>(w);
    float s = 0.0f;
    for (int c = lane; c < nch; c += 64) s = sq8(xr[c], s);
    s = butterfly<64>(s);
    const float rinv = 1.0f / sqrtf(s / (float)H + eps);
    for (int c = lane; c < nch; c += 64) {
        uint4 v = xr[c], g = nw[c], o;
        o.x = (u32)f2bf((bflo(v.x) * rinv) * bflo(g.x)) | ((u32)f2bf((bfhi(v.x) * rinv) * bfhi(g.x)) << 16);
        o.y = (u32)f2bf((bflo(v.y) * rinv) * bflo(g.y)) | ((u32)f2bf((bfhi(v.y) * rinv) * bfhi(g.y)) << 16);
        o.z = (u32)f2bf((bflo(v.z) * rinv) * bflo(g.z)) | ((u32)f2bf((bfhi(v.z) * rinv) * bfhi(g.z)) << 16);
        o.w = (u32)f2bf((bflo(v.w) * rinv) * bflo(g.w)) | ((u32)f2bf((bfhi(v.w) * rinv) * bfhi(g.w)) << 16);
        *reinterpret_cast<uint4*>(y + frag_off(row, c * 8, H)) = o;
    }
}

int vox_launch_rmsnorm(hipStream_t st, const void* x, const void* w, void* y, int rows, int H, float eps) {
    if (H % 8) return vox_fail(VOX_ERR_INVALID, "rmsnorm: cols%8!=0");
    if (rows <= 0) return VOX_OK;
    hipLaunchKernelGGL(k_rmsnorm, dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w,
                       (bf16_t*)y, rows, H, eps);
    return VOX_OK;
}

// ================================================================================================
// head prepare: [per-head RMSNorm] -> RoPE -> q buffer / paged KV append.  One wave per (row, head).
// ================================================================================================
struct HeadArgs {
    const bf16_t *q_src, *k_src, *v_src;  // row strides below (elements)
    long q_stride, k_stride, v_stride;
    bf16_t *q_out, *k_out;                // [N,Hq,D] / [N,Hkv,D] (k_out optional)
    bf16_t* kv;                           // paged cache layer or NULL
    const bf16_t *qn, *kn;
    const float* cs;
    const int *pos, *page, *slot;
    float eps;
    int Hq, Hkv, D, rot, interleave, page_size, table_max_pos;
};

__global__ __launch_bounds__(64) void k_head_prepare(HeadArgs a) {
    __shared__ float sh[512];
    __shared__ __attribute__((aligned(16))) bf16_t so[512];
    const int head = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    const int D = a.D, lpt = D >> 3;
    const int kind = head < a.Hq ? 0 : (head < a.Hq + a.Hkv ? 1 : 2);
    const int h = kind == 0 ? head : (kind == 1 ? head - a.Hq : head - a.Hq - a.Hkv);
    const bf16_t* src = kind == 0 ? a.q_src + (size_t)n * a.q_stride + (size_t)h * D
                       : kind == 1 ? a.k_src + (size_t)n * a.k_stride + (size_t)h * D
                                   : a.v_src + (size_t)n * a.v_stride + (size_t)h * D;
    if (kind == 2 && !a.v_src) return;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (lane < lpt) v = reinterpret_cast<const uint4*>(src)[lane];
    bf16_t* dst = nullptr;
    const size_t ps = (size_t)2 * a.page_size * a.Hkv * D;
    if (kind == 0) dst = a.q_out + ((size_t)n * a.Hq + h) * D;
    else if (a.kv) {
        const int pg = a.page[n];
        if (pg >= 0)
            dst = a.kv + (size_t)pg * ps + ((size_t)(kind == 2 ? a.page_size : 0) + a.slot[n]) * a.Hkv * D + (size_t)h * D;
    } else if (kind == 1 && a.k_out) dst = a.k_out + ((size_t)n * a.Hkv + h) * D;
    if (kind == 2) {
        if (dst && lane < lpt) reinterpret_cast<uint4*>(dst)[lane] = v;
        return;
    }
    float e[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
    const bf16_t* nw = kind == 0 ? a.qn : a.kn;
    if (nw) {
        float s = sq8(v, 0.0f);
        s = butterfly<64>(s);
        const float rinv = 1.0f / sqrtf(s / (float)D + a.eps);
        if (lane < lpt) {
            const uint4 g = reinterpret_cast<const uint4*>(nw)[lane];
            const float gw[8] = {bflo(g.x), bfhi(g.x), bflo(g.y), bfhi(g.y), bflo(g.z), bfhi(g.z), bflo(g.w), bfhi(g.w)};
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = bfround((e[i] * rinv) * gw[i]);
        }
    }
    if (lane < lpt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sh[lane * 8 + i] = e[i];
            so[lane * 8 + i] = f2bf(e[i]);
        }
    }
    __syncthreads();
    if (a.cs) {
        const int half = a.rot >> 1;
        int p = a.pos[n];
        p = p < 0 ? 0 : (p >= a.table_max_pos ? a.table_max_pos - 1 : p);
        const float* t = a.cs + (size_t)p * half * 2;
        for (int i = lane; i < half; i += 64) {
            const int ia = a.interleave ? 2 * i : i, ib = a.interleave ? 2 * i + 1 : i + half;
            const float x = sh[ia], y = sh[ib], c = t[2 * i], s = t[2 * i + 1];
            const float xc = x * c, yc = y * c;
            so[ia] = f2bf(__fmaf_rn(-y, s, xc));
            so[ib] = f2bf(__fmaf_rn(x, s, yc));
        }
        __syncthreads();
    }
    if (dst && lane < lpt) reinterpret_cast<uint4*>(dst)[lane] = reinterpret_cast<const uint4*>(so)[lane];
}

int vox_launch_head_prepare(hipStream_t st, const HeadCall& c) {
    if (c.D % 8 || c.D > 512 || c.N <= 0) return c.N <= 0 ? VOX_OK : vox_fail(VOX_ERR_INVALID, "head_prepare: bad D");
    HeadArgs a{};
    a.q_src = (const bf16_t*)c.q_src; a.k_src = (const bf16_t*)c.k_src; a.v_src = (const bf16_t*)c.v_src;
    a.q_stride = c.q_stride; a.k_stride = c.k_stride; a.v_stride = c.v_stride;
    a.q_out = (bf16_t*)c.q_out; a.k_out = (bf16_t*)c.k_out; a.kv = (bf16_t*)c.kv;
    a.qn = (const bf16_t*)c.qn; a.kn = (const bf16_t*)c.kn; a.cs = c.cs; a.pos = c.pos; a.page = c.page;
    a.slot = c.slot; a.eps = c.eps; a.Hq = c.Hq; a.Hkv = c.Hkv; a.D = c.D; a.rot = c.rot;
    a.interleave = c.interleave; a.page_size = c.page_size; a.table_max_pos = c.table_max_pos;
    const int heads = c.Hq + c.Hkv + (c.v_src ? c.Hkv : 0);
    hipLaunchKernelGGL(k_head_prepare, dim3(heads, c.N), dim3(64), 0, st, a);
    return VOX_OK;
}

#ifdef VOX_DEV_KNOBS
// development builds: phase time stamps (s_memrealtime, 100 MHz) of block (0, 0) of the one-launch decode attention;
// slot 0 of the buffer counts launches, launch i writes stamps [16 * (i + 1) .. ).  Set with vox_dev_set_stamps().
__device__ unsigned long long* g_vox_stamps = nullptr;
extern "C" int vox_dev_set_stamps(void* p) {
    unsigned long long* q = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_vox_stamps), &q, sizeof(q)) == hipSuccess ? 0 : -1;
}
#define VOX_STAMP_DECL unsigned long long* stamp_base = nullptr; \
    if (g_vox_stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { \
        const unsigned long long li = atomicAdd(g_vox_stamps, 1ull); stamp_base = li < 4000 ? g_vox_stamps + 16 * (li + 1) : nullptr; }
#define VOX_STAMP(k) if (stamp_base) stamp_base[k] = wall_clock64();
// the same for the persistent depth step: 32 stamps per launch (block 0, thread 0), buffer set with vox_dev_set_stamps2()
__device__ unsigned long long* g_vox_stamps2 = nullptr;
extern "C" int vox_dev_set_stamps2(void* p) {
    unsigned long long* q = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_vox_stamps2), &q, sizeof(q)) == hipSuccess ? 0 : -1;
}
#define VOX_STAMP2_DECL unsigned long long* stamp2_base = nullptr; \
    if (g_vox_stamps2 && blockIdx.x == 0 && threadIdx.x == 0) { \
        const unsigned long long li = atomicAdd(g_vox_stamps2, 1ull); stamp2_base = li < 2000 ? g_vox_stamps2 + 32 * (li + 1) : nullptr; }
#define VOX_STAMP2(k) if (stamp2_base) stamp2_base[k] = wall_clock64();
#else
#define VOX_STAMP_DECL
#define VOX_STAMP(k)
#define VOX_STAMP2_DECL
#define VOX_STAMP2(k)
#endif

// ================================================================================================
// chunked paged attention: one block per (chunk of 32 KV tokens, kv head, query row)
// ================================================================================================
struct AttnArgs {
    const bf16_t *q, *kv;
    const int *q_req, *q_kvlen, *indptr, *indices;
    float *part_o, *part_ml;
    float scale;
    int Hq, Hkv, page_size, max_chunks;
    int hnd_probe;      // development builds only (VOX_ATTN_HND_PROBE=1)
    int page_shift;     // log2(page_size) when it is a power of two (set_page_size), else -1: token -> (page, slot) by shift / mask instead of
                        // an integer division per K/V element (no divide instruction: ~40 VALU each; 8..16 of them per thread stood in
                        // front of the K/V requests of every decode attention)
    // fused decode mode: q/k/v of each row's own (newest) token come straight from the projection output;
    // per-head norm + RoPE happen here and the block owning the last chunk appends K/V to the cache.
    const bf16_t* qkv;      // [N, (Hq+2Hkv)*D]
    bf16_t* kv_w;           // writable alias of kv
    const bf16_t *qn, *kn;
    const float* cs;
    const int *pos, *page, *slot;
    float eps;
    int rot, interleave, table_max_pos;
    // shortcuts that cut the dependent-load chain in front of the K/V tile loads:
    const int* ptab;     // optional per-row page table [rows][pt_stride] (else indices[indptr[q_req[row]] + j])
    int pt_stride;
    int fixed_kvlen;     // > 0: every row sees exactly this many tokens (depth loop: known on the host)
    int fixed_pos;       // >= 0: every row's RoPE position
    int identity_pages;  // 1: request r owns the single page r and q_req[row] == row (depth loop)
    bf16_t* out;   // single-chunk launches write the final bf16 output here (merge of one chunk == o/l)
    bf16_t* out_frag;   // optional fragment-major copy of the output (see LinearCall)
    int hoist;          // 1: page ids requested before the row length is known (k_attn_decode8 with a per-row page table)
};
static inline void set_page_size(AttnArgs& a, int page_size) {
    a.page_size = page_size;
    a.page_shift = -1;
    if (page_size > 0 && (page_size & (page_size - 1)) == 0) { a.page_shift = 0; while ((1 << a.page_shift) < page_size) ++a.page_shift; }
}
__device__ __forceinline__ int page_of(const AttnArgs& a, int tok) { return a.page_shift >= 0 ? tok >> a.page_shift : tok / a.page_size; }
__device__ __forceinline__ int slot_of(const AttnArgs& a, int tok) { return a.page_shift >= 0 ? tok & (a.page_size - 1) : tok % a.page_size; }

// norm (optional) + rope of one head held as one 16-byte chunk per lane (lanes < LPT); result as bf16 bits in
// `out` (LDS, D elements).  sh: LDS scratch of D floats.  All 64 lanes of the wave must call this.
struct PrepOps { u32x4_t v, g; float csx, csy; };
// the three operands (head row, norm weight, this lane's RoPE table entry) are requested together: one exposed L2 latency instead
// of three dependent ones.  Branch-free (every lane requests something in range, prep_head_apply selects): a conditional load
// would be joined with its default by a copy behind a wait, in front of whatever the caller requests next.
template <int D>
__device__ __forceinline__ PrepOps prep_head_load(const bf16_t* src, const bf16_t* nw, const float* cs_row, int rot, int lane) {
    constexpr int LPT = D / 8;
    PrepOps o;
    const int l = lane < LPT ? lane : 0;
    o.v = reinterpret_cast<const u32x4_t*>(src)[l];
    o.g = reinterpret_cast<const u32x4_t*>(nw ? nw : src)[l];
    const float2 c = *reinterpret_cast<const float2*>((cs_row ? cs_row : reinterpret_cast<const float*>(src)) + 2 * (lane < (rot >> 1) ? lane : 0));
    o.csx = c.x; o.csy = c.y;
    return o;
}
// ... the head row staged in LDS by the caller (q | k | v of the new token gathered from hand-off granules: k_talker_layers); norm weight and
// RoPE table must exist (no pointer selects between LDS and global memory)
template <int D>
__device__ __forceinline__ PrepOps prep_head_load_lds(const bf16_t* src_lds, const bf16_t* nw, const float* cs_row, int rot, int lane) {
    constexpr int LPT = D / 8;
    PrepOps o;
    const int l = lane < LPT ? lane : 0;
    o.v = reinterpret_cast<const u32x4_t*>(src_lds)[l];
    o.g = *((const VOX_GLOBAL_AS u32x4_t*)nw + l);
    const VOX_GLOBAL_AS float* cp = (const VOX_GLOBAL_AS float*)cs_row + 2 * (lane < (rot >> 1) ? lane : 0);
    o.csx = cp[0]; o.csy = cp[1];
    return o;
}
// the operands are in registers from here on (the caller's own wait may already have covered them)
__device__ __forceinline__ void prep_head_arrived(PrepOps& o) { asm volatile("" : "+v"(o.v), "+v"(o.g), "+v"(o.csx), "+v"(o.csy)); }
template <int D>
__device__ __forceinline__ void prep_head_apply(const PrepOps& o, const bf16_t* nw, float eps, const float* cs_row,
                                                int rot, int interleave, float* sh, bf16_t* out, int lane) {
    constexpr int LPT = D / 8;
    const int half = rot >> 1;
    const bool own = lane < LPT;
    const uint4 v = own ? as_uint4(o.v) : make_uint4(0, 0, 0, 0), g = own && nw ? as_uint4(o.g) : make_uint4(0, 0, 0, 0);
    const float2 cs0 = cs_row && lane < half ? make_float2(o.csx, o.csy) : make_float2(1.0f, 0.0f);
    float e[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
    if (nw) {
        float s = sq8(v, 0.0f);
        s = butterfly<64>(s);
        const float rinv = 1.0f / sqrtf(s / (float)D + eps);
        if (lane < LPT) {
            const float gw[8] = {bflo(g.x), bfhi(g.x), bflo(g.y), bfhi(g.y), bflo(g.z), bfhi(g.z), bflo(g.w), bfhi(g.w)};
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = bfround((e[i] * rinv) * gw[i]);
        }
    }
    if (lane < LPT) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sh[lane * 8 + i] = e[i];
            out[lane * 8 + i] = f2bf(e[i]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (cs_row) {
        for (int i = lane; i < half; i += 64) {
            const int ia = interleave ? 2 * i : i, ib = interleave ? 2 * i + 1 : i + half;
            const float x = sh[ia], y = sh[ib], c = i == lane ? cs0.x : cs_row[2 * i], sn = i == lane ? cs0.y : cs_row[2 * i + 1];
            const float xc = x * c, yc = y * c;
            out[ia] = f2bf(__fmaf_rn(-y, sn, xc));
            out[ib] = f2bf(__fmaf_rn(x, sn, yc));
        }
    }
}
template <int D>
__device__ __forceinline__ void prep_head(const bf16_t* src, const bf16_t* nw, float eps, const float* cs_row,
                                          int rot, int interleave, float* sh, bf16_t* out, int lane) {
    const PrepOps o = prep_head_load<D>(src, nw, cs_row, rot, lane);
    prep_head_apply<D>(o, nw, eps, cs_row, rot, interleave, sh, out, lane);
}

template <int D, bool FUSED>
__global__ __launch_bounds__(256) void k_attn_partial(AttnArgs a) {
    constexpr int LPT = D / 8;        // lanes per token
    constexpr int TPW = 64 / LPT;     // tokens per wave pass
    constexpr int GMAX = 16;
    __shared__ __attribute__((aligned(16))) uint4 Ks[VOX_TC * LPT];
    __shared__ __attribute__((aligned(16))) uint4 Vs[VOX_TC * LPT];
    __shared__ __attribute__((aligned(16))) uint4 Qs[GMAX * LPT];
    __shared__ float S[GMAX][VOX_TC];
    __shared__ float Ms[GMAX];
    __shared__ float Sh[FUSED ? 4 * D : 1];
    __shared__ __attribute__((aligned(16))) bf16_t Knew[FUSED ? D : 8];

    const int c = blockIdx.x, hk = blockIdx.y, row = blockIdx.z;
    const int L = a.fixed_kvlen > 0 ? a.fixed_kvlen : a.q_kvlen[row];
    // (fused form) where the row's new token is appended, and its V row: requested with the row length instead of behind the tile's
    // barrier, where they were three dependent round trips (page, slot, V row) in front of the scores
    int pg_pre = row, sl_pre = 0;
    uint4 vx_pre = make_uint4(0, 0, 0, 0);
    if (FUSED) {
        if (!a.identity_pages) { pg_pre = a.page[row]; sl_pre = a.slot[row]; }
        vx_pre = reinterpret_cast<const uint4*>(a.qkv + (size_t)row * (a.Hq + 2 * a.Hkv) * D + (size_t)(a.Hq + a.Hkv) * D + (size_t)hk * D)[threadIdx.x % LPT];
    }
    const int t0 = c * VOX_TC;
    if (t0 >= L) return;
    const int nt = (L - t0) < VOX_TC ? (L - t0) : VOX_TC;
    const int G = a.Hq / a.Hkv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* pages = a.identity_pages ? nullptr
                       : (a.ptab ? a.ptab + (size_t)row * a.pt_stride : a.indices + a.indptr[a.q_req[row]]);
    const size_t ps = (size_t)2 * a.page_size * a.Hkv * D;
    const bool own_last = FUSED && (t0 + nt == L);   // this chunk holds the row's newest token (index L-1)

    // the K/V tile is requested first and parked in LDS only after the q prologue below: its HBM/L2 latency overlaps the
    // prologue's own dependent loads (q row, norm weight, RoPE table) instead of preceding them
    constexpr int KVL = (VOX_TC * LPT + 255) / 256;
    uint4 kreg[KVL], vreg[KVL];
#pragma unroll
    for (int u = 0; u < KVL; ++u) {
        const int i = tid + 256 * u, t = i / LPT, j = i % LPT;
        kreg[u] = make_uint4(0, 0, 0, 0);
        vreg[u] = kreg[u];
        if (i < VOX_TC * LPT && t < nt && !(own_last && t == nt - 1)) {
            const int tok = t0 + t;
            const int pgi = pages ? pages[page_of(a, tok)] : row;
            const bf16_t* base = a.kv + (size_t)pgi * ps + ((size_t)slot_of(a, tok) * a.Hkv + hk) * D;
            kreg[u] = reinterpret_cast<const uint4*>(base)[j];
            vreg[u] = reinterpret_cast<const uint4*>(base + (size_t)a.page_size * a.Hkv * D)[j];
        }
    }
    if (!FUSED) {
        for (int i = tid; i < G * LPT; i += 256) {
            const int g = i / LPT, j = i % LPT;
            Qs[i] = reinterpret_cast<const uint4*>(a.q + ((size_t)row * a.Hq + hk * G + g) * D)[j];
        }
    } else {
        const int nqkv = (a.Hq + 2 * a.Hkv) * D;
        const bf16_t* raw = a.qkv + (size_t)row * nqkv;
        int p = a.fixed_pos >= 0 ? a.fixed_pos : a.pos[row];
        p = p < 0 ? 0 : (p >= a.table_max_pos ? a.table_max_pos - 1 : p);
        const float* cs_row = a.cs ? a.cs + (size_t)p * (a.rot >> 1) * 2 : nullptr;
        // heads 0..G-1: q heads of this kv head; head G: the new k (only where needed)
        const int nh = G + (own_last ? 1 : 0);
        for (int h = wave; h < nh; h += 4) {
            const bool isk = h == G;
            const bf16_t* src = isk ? raw + (size_t)a.Hq * D + (size_t)hk * D : raw + (size_t)(hk * G + h) * D;
            bf16_t* dst = isk ? Knew : reinterpret_cast<bf16_t*>(Qs) + (size_t)h * D;
            prep_head<D>(src, isk ? a.kn : a.qn, a.eps, cs_row, a.rot, a.interleave, Sh + wave * D, dst, lane);
        }
    }
#pragma unroll
    for (int u = 0; u < KVL; ++u) {
        const int i = tid + 256 * u;
        if (i < VOX_TC * LPT) { Ks[i] = kreg[u]; Vs[i] = vreg[u]; }
    }
    __syncthreads();
    if (own_last) {
        // place the new token into the tile and append it to the paged cache (page < 0: graph padding row)
        const int pg = pg_pre;
        const int sl = a.identity_pages ? (L - 1) : sl_pre;
        if (tid < LPT) {
            const uint4 kx = reinterpret_cast<const uint4*>(Knew)[tid];
            const uint4 vx = vx_pre;
            Ks[(nt - 1) * LPT + tid] = kx;
            Vs[(nt - 1) * LPT + tid] = vx;
            if (pg >= 0) {
                bf16_t* base = a.kv_w + (size_t)pg * ps + ((size_t)sl * a.Hkv + hk) * D;
                reinterpret_cast<uint4*>(base)[tid] = kx;
                reinterpret_cast<uint4*>(base + (size_t)a.page_size * a.Hkv * D)[tid] = vx;
            }
        }
        __syncthreads();
    }

    // scores: LPT lanes per token, butterfly over LPT lanes
    for (int tb = wave * TPW; tb < VOX_TC; tb += 4 * TPW) {
        const int tt = tb + lane / LPT, j = lane % LPT;
        const uint4 kx = Ks[tt * LPT + j];
        for (int g = 0; g < G; ++g) {
            float d = dot8(Qs[g * LPT + j], kx, 0.0f);
            d = butterfly<LPT>(d);
            if (j == 0) S[g][tt] = d * a.scale;
        }
    }
    __syncthreads();
    // chunk max + p = exp2((s-m)*log2e): 32 lanes per q head
    for (int pr = tid; pr < G * VOX_TC; pr += 256) {
        const int g = pr / VOX_TC, t = pr % VOX_TC;
        const float s = t < nt ? S[g][t] : -INFINITY;
        float m = s;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, VOX_WAVE));
        const float p = t < nt ? exp2_c((s - m) * VOX_LOG2E) : 0.0f;
        S[g][t] = p;
        if (t == 0) Ms[g] = m;
    }
    __syncthreads();
    // PV: one thread per (q head, pair of dims); sequential over the tokens of the zero-padded tile (p = 0 and V = 0 beyond the chunk's last
    // token: l + 0 and fma(0, 0, o) leave the sums bit-unchanged), so the loop has no trip count to wait for and reads 32-bit words
    const u32* Vw = reinterpret_cast<const u32*>(Vs);
    for (int e = tid; e < G * (D / 2); e += 256) {
        const int g = e / (D / 2), dp = e % (D / 2);
        float o0 = 0.0f, o1 = 0.0f, l = 0.0f;
#pragma unroll
        for (int t = 0; t < VOX_TC; ++t) {
            const float p = S[g][t];
            const u32 vw = Vw[t * (D / 2) + dp];
            l = l + p;
            o0 = __fmaf_rn(p, bflo(vw), o0);
            o1 = __fmaf_rn(p, bfhi(vw), o1);
        }
        const size_t hi = (size_t)row * a.Hq + hk * G + g;
        const int d = 2 * dp;
        if (a.out) {   // one chunk: w = exp2(0) = 1, O = fma(o,1,0) = o, L = l
            const bf16_t r0 = f2bf(o0 / l), r1 = f2bf(o1 / l);
            *reinterpret_cast<u32*>(a.out + hi * D + d) = (u32)r0 | ((u32)r1 << 16);
            if (a.out_frag) {
                a.out_frag[frag_off(row, (hk * G + g) * D + d, a.Hq * D)] = r0;
                a.out_frag[frag_off(row, (hk * G + g) * D + d + 1, a.Hq * D)] = r1;
            }
            continue;
        }
        *reinterpret_cast<float2*>(a.part_o + (hi * a.max_chunks + c) * D + d) = make_float2(o0, o1);
        if (dp == 0) {
            a.part_ml[(hi * a.max_chunks + c) * 2 + 0] = Ms[g];
            a.part_ml[(hi * a.max_chunks + c) * 2 + 1] = l;
        }
    }
}

int vox_launch_attn_partial(hipStream_t st, const AttnCall& c) {
    if (c.Nq <= 0) return VOX_OK;
    if (c.Hq % c.Hkv || c.Hq / c.Hkv > 16) return vox_fail(VOX_ERR_INVALID, "attention: unsupported GQA group");
    AttnArgs a{};
    a.q = (const bf16_t*)c.q; a.kv = (const bf16_t*)c.kv; a.q_req = c.q_req; a.q_kvlen = c.q_kvlen;
    a.indptr = c.indptr; a.indices = c.indices; a.part_o = c.part_o; a.part_ml = c.part_ml; a.scale = c.scale;
    a.Hq = c.Hq; a.Hkv = c.Hkv; set_page_size(a, c.page_size); a.max_chunks = c.max_chunks;
    a.qkv = (const bf16_t*)c.qkv; a.kv_w = (bf16_t*)const_cast<void*>(c.kv); a.qn = (const bf16_t*)c.qn;
    a.kn = (const bf16_t*)c.kn; a.cs = c.cs; a.pos = c.pos; a.page = c.page; a.slot = c.slot; a.eps = c.eps;
    a.rot = c.rot; a.interleave = c.interleave; a.table_max_pos = c.table_max_pos;
    a.ptab = c.ptab; a.pt_stride = c.pt_stride; a.fixed_kvlen = c.fixed_kvlen; a.fixed_pos = c.fixed_pos;
    a.identity_pages = c.identity_pages;
    a.out = nullptr;
    int nchunk = (c.max_kvlen + VOX_TC - 1) / VOX_TC;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > c.max_chunks) return vox_fail(VOX_ERR_INVALID, "attention: max_kvlen exceeds workspace");
    if (nchunk == 1) { a.out = (bf16_t*)c.out; a.out_frag = (bf16_t*)c.out_frag; }
    dim3 grid(nchunk, c.Hkv, c.Nq);
    const bool fused = c.qkv != nullptr;
#define VOX_ATT(D_)                                                                           \
    if (c.D == D_) {                                                                          \
        if (fused) hipLaunchKernelGGL((k_attn_partial<D_, true>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((k_attn_partial<D_, false>), grid, dim3(256), 0, st, a);      \
        return VOX_OK;                                                                        \
    }
    VOX_ATT(128) VOX_ATT(64) VOX_ATT(16)
#undef VOX_ATT
    return vox_fail(VOX_ERR_INVALID, "attention: head_dim must be 16, 64 or 128");
}

// The tile's requests are written as asm: as plain loads the compiler keeps the zero fill and the conditional load in different
// registers and joins them with a copy after EACH load (s_waitcnt vmcnt(0) + v_mov per request: one memory round trip per request
// instead of one per tile).  Here the destination is pre-zeroed and tied ("+v"): lanes without a token are masked out by the
// branch and keep the zeros, and all requests of the tile go out back to back.  The compiler does not know these registers are in
// flight (a live-range split between request and wait would copy stale data), so kv_wait() follows the requests DIRECTLY: the
// first head's q / k operands are requested in front of the tile (they were behind it and returned after it anyway — loads
// return in order), and only the first tile is fetched this way (a second tile is requested under the first one's arithmetic,
// off the critical path, as plain loads).
#define VOX_KV_ASM_FETCH \
    u32x4_t kreg[KVL], vreg[KVL]; \
    if (!hoist) { \
_Pragma("unroll") \
        for (int ci = 0; ci < CPG; ++ci) \
_Pragma("unroll") \
            for (int u = 0; u < KVL; ++u) { \
                const int tok = (grp + NG * ci) * VOX_TC + (gt + GT * u) / LPT; \
                if (one_page && u > 0) { pgi_pre[ci][u] = 0; continue; } \
                pgi_pre[ci][u] = pages ? (tok < L - 1 ? pages[page_of(a, tok)] : 0) : row; \
            } \
    } \
    auto pages_arrived = [&]() { /* the first tile's page ids are in registers (no waits between its requests) */ \
_Pragma("unroll") \
        for (int u = 0; u < KVL; ++u) asm volatile("" : "+v"(pgi_pre[0][u])); \
    }; \
    auto fetch_tile = [&](int ci) { \
        const int t0 = (grp + NG * ci) * VOX_TC; \
_Pragma("unroll") \
        for (int u = 0; u < KVL; ++u) { \
            kreg[u] = u32x4_t{0u, 0u, 0u, 0u}; \
            vreg[u] = u32x4_t{0u, 0u, 0u, 0u}; \
        } \
        /* A 32-token chunk lies inside ONE page when the page size is a power of two >= 32 (one_page): the thread's requests are then */ \
        /* base + u * stride — one 64-bit address computation per tile instead of one per request (page select, shift / mask, two */ \
        /* 64-bit multiply-adds: ~85 instructions between two requests of a wave that has 8 .. 16 of them to issue) */ \
        const bf16_t* fast_base = nullptr; \
        size_t fast_stride = 0; \
        if (one_page) { \
            fast_base = a.kv + (size_t)pgi_pre[ci][0] * ps + ((size_t)((t0 & (a.page_size - 1)) + gt / LPT) * a.Hkv + hk) * D; \
            fast_stride = (size_t)(GT / LPT) * a.Hkv * D; \
        } \
_Pragma("unroll") \
        for (int u = 0; u < KVL; ++u) { \
            const int i = gt + GT * u, t = i / LPT, j = i % LPT; \
            const int tok = t0 + t; \
            if (i < VOX_TC * LPT && tok < L - 1) { \
                const int pgi = pgi_pre[ci][one_page ? 0 : u]; \
                const bf16_t* base = one_page ? fast_base + (size_t)u * fast_stride : a.kv + (size_t)pgi * ps + ((size_t)slot_of(a, tok) * a.Hkv + hk) * D; \
                const uint4* kp = reinterpret_cast<const uint4*>(base) + j; \
                const uint4* vp = reinterpret_cast<const uint4*>(base + (size_t)a.page_size * a.Hkv * D) + j; \
                if (ci == 0) { \
                    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(kreg[u]) : "v"(kp) : "memory"); \
                    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(vreg[u]) : "v"(vp) : "memory"); \
                } else { \
                    kreg[u] = *reinterpret_cast<const u32x4_t*>(kp); \
                    vreg[u] = *reinterpret_cast<const u32x4_t*>(vp); \
                } \
            } else if (!kRawLds && i < VOX_TC * LPT && tok == L - 1) { \
                const uint4* vp = reinterpret_cast<const uint4*>(raw + (size_t)(a.Hq + a.Hkv) * D + (size_t)hk * D) + j; \
                if (ci == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(vreg[u]) : "v"(vp) : "memory"); \
                else vreg[u] = *reinterpret_cast<const u32x4_t*>(vp); \
            } \
        } \
    }; \
    auto kv_wait = [&]() { \
        static_assert(KVL == 4 || KVL == 8 || KVL == 2 || KVL == 1, "kv_wait lists the tile registers"); \
        if constexpr (KVL == 8) \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(kreg[0]), "+v"(kreg[1]), "+v"(kreg[2]), "+v"(kreg[3]), "+v"(kreg[4]), "+v"(kreg[5]), "+v"(kreg[6]), "+v"(kreg[7]), \
                         "+v"(vreg[0]), "+v"(vreg[1]), "+v"(vreg[2]), "+v"(vreg[3]), "+v"(vreg[4]), "+v"(vreg[5]), "+v"(vreg[6]), "+v"(vreg[7]) :: "memory"); \
        else if constexpr (KVL == 4) \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(kreg[0]), "+v"(kreg[1]), "+v"(kreg[2]), "+v"(kreg[3]), \
                         "+v"(vreg[0]), "+v"(vreg[1]), "+v"(vreg[2]), "+v"(vreg[3]) :: "memory"); \
        else if constexpr (KVL == 2) \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(kreg[0]), "+v"(kreg[1]), "+v"(vreg[0]), "+v"(vreg[1]) :: "memory"); \
        else \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(kreg[0]), "+v"(vreg[0]) :: "memory"); \
    };

// ================================================================================================
// One-launch decode attention for contexts of up to 8 chunks (<= 256 visible tokens): one 1024-thread block per
// (kv head, row) runs every chunk of that pair — NG groups of 1024 / NG threads, group q taking chunks q, q + NG, ... — and
// merges the partials in LDS: the separate merge launch (a whole dependent stage of the frame: 4.7 us + a kernel boundary
// per talker layer) disappears.  Per chunk the arithmetic is k_attn_partial<D, true>'s, the merge is k_attn_merge's (global
// max, then L / O accumulated over the chunks in ascending order): bit-identical outputs.  With NG = 8 every chunk has its own
// group of two waves (all chunks in flight at once); K/V tiles are requested before the q prologue.  The token loops run
// over the whole zero-padded tile without predicates (p = 0 beyond the chunk's last token: l + 0 and fma(0, v, o) leave the
// sums bit-unchanged), so that the LDS reads of consecutive tokens overlap.
// (An earlier one-launch variant gave each chunk ONE wave and lost: a 32-token chunk is too long a serial chain for a wave.)
template <int D, int GMAX, int NG, int NCH = 8>
__global__ __launch_bounds__(1024) void k_attn_decode8(AttnArgs a) {
    // NCH = 16 (CPG = 2 chunks per group): contexts of up to 512 tokens — the second chunk's tile is requested when the first one has been
    // parked, so its HBM latency runs under the first chunk's scores / softmax / P.V (one tile's registers at a time)
    constexpr int LPT = D / 8, TPW = 64 / LPT, GT = 1024 / NG, GW = GT / 64, CPG = NCH / NG;
    constexpr int KVL = (VOX_TC * LPT + GT - 1) / GT;
    constexpr bool QREG = GMAX <= 2;      // q heads held unpacked in registers during the score passes
    __shared__ __attribute__((aligned(16))) uint4 Ks[NG][VOX_TC * LPT];
    __shared__ __attribute__((aligned(16))) uint4 Vs[NG][VOX_TC * LPT];
    __shared__ __attribute__((aligned(16))) uint4 Qs[GMAX * LPT];
    __shared__ float S[NG][GMAX][VOX_TC];
    __shared__ float Ms[NG][GMAX];
    __shared__ float Sh[16 * D];
    __shared__ __attribute__((aligned(16))) bf16_t Knew[D];
    __shared__ float Po[NCH][GMAX][D];
    __shared__ float2 Pml[NCH][GMAX];
    __shared__ float Wm[NCH][GMAX];

    // head split: gridDim.x = Hkv * HS blocks per row, block (hk, hs) takes q heads hs * G .. hs * G + G - 1 of kv head hk's group
    const int HS = gridDim.x / a.Hkv, hk = blockIdx.x / HS, hs = blockIdx.x % HS, row = blockIdx.y;
    VOX_STAMP_DECL
    VOX_STAMP(0)
    VOX_TR_DECL
    const int tid = threadIdx.x, lane = tid & 63, wave16 = tid >> 6;
    const int grp = tid / GT, gt = tid % GT, gw = wave16 % GW;
    const int* pages = a.identity_pages ? nullptr
                       : (a.ptab ? a.ptab + (size_t)row * a.pt_stride : a.indices + a.indptr[a.q_req[row]]);
    // per-row page table: the page ids of this thread's tokens do not depend on the row's length, so they are requested
    // together with it (one exposed round trip in front of the K/V loads instead of two); entries past the row's last page are
    // read (clamped to the table row) and never used
    int pgi_pre[CPG][KVL];
    const bool hoist = a.ptab && a.hoist;
    const bool one_page = a.page_shift >= 5 && (GT % LPT) == 0;      // (VOX_TC = 32 tokens: a chunk never straddles a page; VOX_KV_ASM_FETCH)
    if (hoist) {
#pragma unroll
        for (int ci = 0; ci < CPG; ++ci)
#pragma unroll
            for (int u = 0; u < KVL; ++u) {
                if (one_page && u > 0) { pgi_pre[ci][u] = 0; continue; }      // one page id per chunk
                const int tok = (grp + NG * ci) * VOX_TC + (gt + GT * u) / LPT;
                const int pi = page_of(a, tok);
                pgi_pre[ci][u] = pages[pi < a.pt_stride ? pi : a.pt_stride - 1];
            }
    }
    const int L = a.fixed_kvlen > 0 ? a.fixed_kvlen : a.q_kvlen[row];
    // (the new token's page / slot words with the row length: one exposed round trip less in front of the append — attn_decode8_block)
    const int pg_new = a.identity_pages ? row : a.page[row];
    const int sl_new = a.identity_pages ? 0 : a.slot[row];
    const int nc = (L + VOX_TC - 1) / VOX_TC;                 // 1..NCH
    const int Gf = a.Hq / a.Hkv, G = Gf / HS, g0 = hs * G;
    const size_t ps = (size_t)2 * a.page_size * a.Hkv * D;
    const int nqkv = (a.Hq + 2 * a.Hkv) * D;
    const bf16_t* raw = a.qkv + (size_t)row * nqkv;

    // K/V tiles of this group's chunks.  The row's newest token (index L - 1) comes from the projection output: its V row is
    // loaded into the tile here, its K row (per-head norm + RoPE below) is read from Knew by the score pass.
#if VOX_KV_ASM
    constexpr bool kRawLds = false;
    VOX_KV_ASM_FETCH
#else
    uint4 kreg[KVL], vreg[KVL];
    auto fetch_tile = [&](int ci) {
        const int t0 = (grp + NG * ci) * VOX_TC;
#pragma unroll
        for (int u = 0; u < KVL; ++u) {
            const int i = gt + GT * u, t = i / LPT, j = i % LPT;
            kreg[u] = make_uint4(0, 0, 0, 0);
            vreg[u] = kreg[u];
            const int tok = t0 + t;
            if (i < VOX_TC * LPT && tok < L - 1) {
                const int pgi = hoist ? pgi_pre[ci][u] : (pages ? pages[page_of(a, tok)] : row);
#ifdef VOX_DEV_KNOBS      // timing probe (wrong data): the tile read as if the page were laid out [head][slot][D] — contiguous 32 KB per head
                const bf16_t* base = a.kv + (size_t)pgi * ps + (a.hnd_probe ? ((size_t)hk * a.page_size + slot_of(a, tok)) * D : ((size_t)slot_of(a, tok) * a.Hkv + hk) * D);
#else
                const bf16_t* base = a.kv + (size_t)pgi * ps + ((size_t)slot_of(a, tok) * a.Hkv + hk) * D;
#endif
                kreg[u] = reinterpret_cast<const uint4*>(base)[j];
                vreg[u] = reinterpret_cast<const uint4*>(base + (size_t)a.page_size * a.Hkv * D)[j];
            } else if (i < VOX_TC * LPT && tok == L - 1) {
                vreg[u] = reinterpret_cast<const uint4*>(raw + (size_t)(a.Hq + a.Hkv) * D + (size_t)hk * D)[j];
            }
        }
    };
    auto kv_wait = [&]() {};
    auto pages_arrived = [&]() {};
#endif
    // q heads of this kv head (per-head norm + RoPE) and the new k: one head per wave; the first head's operands are requested in
    // front of the tile
    int p = a.fixed_pos >= 0 ? a.fixed_pos : a.pos[row];
    p = p < 0 ? 0 : (p >= a.table_max_pos ? a.table_max_pos - 1 : p);
    const float* cs_row = a.cs ? a.cs + (size_t)p * (a.rot >> 1) * 2 : nullptr;
    auto head_src = [&](int h) { return h == G ? raw + (size_t)a.Hq * D + (size_t)hk * D : raw + (size_t)(hk * Gf + g0 + h) * D; };
    PrepOps po0;
    pages_arrived();
    if (wave16 < G + 1) po0 = prep_head_load<D>(head_src(wave16), wave16 == G ? a.kn : a.qn, cs_row, a.rot, lane);
    fetch_tile(0);
    kv_wait();
    if (wave16 < G + 1) prep_head_arrived(po0);
    VOX_STAMP(1) VOX_TR(1)
    for (int h = wave16; h < G + 1; h += 16) {
        const bool isk = h == G;
        bf16_t* dst = isk ? Knew : reinterpret_cast<bf16_t*>(Qs) + (size_t)h * D;
        if (h == wave16) prep_head_apply<D>(po0, isk ? a.kn : a.qn, a.eps, cs_row, a.rot, a.interleave, Sh + wave16 * D, dst, lane);
        else prep_head<D>(head_src(h), isk ? a.kn : a.qn, a.eps, cs_row, a.rot, a.interleave, Sh + wave16 * D, dst, lane);
    }
    VOX_STAMP(2)
#pragma unroll
    for (int ci = 0; ci < CPG; ++ci) {
        const int c = grp + NG * ci, t0 = c * VOX_TC;
        const bool live = t0 < L;
        const int nt = live ? ((L - t0) < VOX_TC ? (L - t0) : VOX_TC) : 0;
        const bool own_last = live && (t0 + nt == L);
        if (ci > 0) __syncthreads();           // the previous chunk's tile is dead
#pragma unroll
        for (int u = 0; u < KVL; ++u) {
            const int i = gt + GT * u;
            if (i < VOX_TC * LPT) { Ks[grp][i] = as_uint4(kreg[u]); Vs[grp][i] = as_uint4(vreg[u]); }
        }
        if (ci + 1 < CPG) fetch_tile(ci + 1);  // in flight during this chunk's arithmetic
        __syncthreads();                       // tiles parked; (ci = 0) Qs / Knew written
        VOX_STAMP(3) VOX_TR(3)
        if (own_last && hs == 0 && gt < LPT) {
            // append the new token to the paged cache (page < 0: graph padding row)
            const int pg = pg_new;
            const int sl = a.identity_pages ? (L - 1) : sl_new;
            if (pg >= 0) {
                bf16_t* base = a.kv_w + (size_t)pg * ps + ((size_t)sl * a.Hkv + hk) * D;
                reinterpret_cast<uint4*>(base)[gt] = reinterpret_cast<const uint4*>(Knew)[gt];
                reinterpret_cast<uint4*>(base + (size_t)a.page_size * a.Hkv * D)[gt] = Vs[grp][(nt - 1) * LPT + gt];
            }
        }
        VOX_STAMP(4)
        if (live) {      // scores: LPT lanes per token, butterfly over LPT lanes; the token's K chunk is unpacked once for all q heads
            const int j = lane % LPT;
            float qf[QREG ? GMAX : 1][8];
            if constexpr (QREG) {
#pragma unroll
                for (int g = 0; g < GMAX; ++g) {
                    const uint4 qx = Qs[(g < G ? g : 0) * LPT + j];
                    qf[g][0] = bflo(qx.x); qf[g][1] = bfhi(qx.x); qf[g][2] = bflo(qx.y); qf[g][3] = bfhi(qx.y);
                    qf[g][4] = bflo(qx.z); qf[g][5] = bfhi(qx.z); qf[g][6] = bflo(qx.w); qf[g][7] = bfhi(qx.w);
                }
            }
#pragma unroll
            for (int tb = gw * TPW; tb < VOX_TC; tb += GW * TPW) {
                const int tt = tb + lane / LPT;
                const uint4 kx = (own_last && tt == nt - 1) ? reinterpret_cast<const uint4*>(Knew)[j] : Ks[grp][tt * LPT + j];
                if constexpr (QREG) {
                    const float kf[8] = {bflo(kx.x), bfhi(kx.x), bflo(kx.y), bfhi(kx.y), bflo(kx.z), bfhi(kx.z), bflo(kx.w), bfhi(kx.w)};
#pragma unroll
                    for (int g = 0; g < GMAX; ++g) {
                        float d = 0.0f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) d = __fmaf_rn(qf[g][e], kf[e], d);
                        d = butterfly<LPT>(d);
                        if (j == 0 && g < G) S[grp][g][tt] = d * a.scale;
                    }
                } else {
                    for (int g = 0; g < G; ++g) {
                        float d = dot8(Qs[g * LPT + j], kx, 0.0f);
                        d = butterfly<LPT>(d);
                        if (j == 0) S[grp][g][tt] = d * a.scale;
                    }
                }
            }
        }
        __syncthreads();
        VOX_STAMP(5) VOX_TR(5)
        if (live) {      // chunk max + p = exp2((s-m)*log2e): 32 lanes per q head
            for (int pr = gt; pr < G * VOX_TC; pr += GT) {
                const int g = pr / VOX_TC, t = pr % VOX_TC;
                const float s = t < nt ? S[grp][g][t] : -INFINITY;
                float m = s;
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, VOX_WAVE));
                const float p = t < nt ? exp2_c((s - m) * VOX_LOG2E) : 0.0f;
                S[grp][g][t] = p;
                if (t == 0) Ms[grp][g] = m;
            }
        }
        __syncthreads();
        VOX_STAMP(6)
        if (live) {      // PV: one thread per (q head, pair of dims); sequential over the tokens of the (zero-padded) tile
            const u32* Vw = reinterpret_cast<const u32*>(Vs[grp]);
            for (int e = gt; e < G * (D / 2); e += GT) {
                const int g = e / (D / 2), dp = e % (D / 2);
                float o0 = 0.0f, o1 = 0.0f, l = 0.0f;
#pragma unroll(NCH == 8 ? VOX_TC : 8)                      // (beside an in-flight tile the full unroll's operand registers would spill)
                for (int t = 0; t < VOX_TC; ++t) {
                    const float p = S[grp][g][t];          // 0 for t >= nt
                    const u32 vw = Vw[t * (D / 2) + dp];
                    l = l + p;
                    o0 = __fmaf_rn(p, bflo(vw), o0);
                    o1 = __fmaf_rn(p, bfhi(vw), o1);
                }
                *reinterpret_cast<float2*>(&Po[c][g][2 * dp]) = make_float2(o0, o1);
                if (dp == 0) Pml[c][g] = make_float2(Ms[grp][g], l);
            }
        }
    }
    __syncthreads();
    VOX_STAMP(7) VOX_TR(7)
    // merge (k_attn_merge): global max, the chunk weights w_c = exp2((m_c - M) log2e) once per (chunk, head), then L and O over
    // the chunks in ascending order
    if (tid < NCH * GMAX) {
        const int c = tid / GMAX, g = tid % GMAX;
        if (g < G) {
            float M = -INFINITY;
            for (int cc = 0; cc < nc; ++cc) M = fmaxf(M, Pml[cc][g].x);
            Wm[c][g] = c < nc ? exp2_c((Pml[c][g].x - M) * VOX_LOG2E) : 0.0f;
        }
    }
    __syncthreads();
    for (int e = tid; e < G * D; e += 1024) {
        const int g = e / D, d = e % D;
        float Lsum = 0.0f, O = 0.0f;
        for (int c = 0; c < nc; ++c) {
            const float w = Wm[c][g];
            Lsum = __fmaf_rn(Pml[c][g].y, w, Lsum);
            O = __fmaf_rn(Po[c][g][d], w, O);
        }
        const bf16_t r = f2bf(O / Lsum);
        const int h = hk * Gf + g0 + g;
        a.out[((size_t)row * a.Hq + h) * D + d] = r;
        if (a.out_frag) a.out_frag[frag_off(row, h * D + d, a.Hq * D)] = r;
    }
    VOX_STAMP(8)
    VOX_TR_END(3, (GMAX << 8) | NCH)
}

#ifdef VOX_DEV_KNOBS      // the previous form of the kernel, for A/B timing in development builds (VOX_ATTN_V1=1)
template <int D, int GMAX, int NG>
__global__ __launch_bounds__(1024) void k_attn_decode8_v1(AttnArgs a) {
    constexpr int LPT = D / 8, TPW = 64 / LPT, NCH = 8, GT = 1024 / NG, GW = GT / 64, CPG = NCH / NG;
    constexpr int KVL = (VOX_TC * LPT + GT - 1) / GT;
    __shared__ __attribute__((aligned(16))) uint4 Ks[NG][VOX_TC * LPT];
    __shared__ __attribute__((aligned(16))) uint4 Vs[NG][VOX_TC * LPT];
    __shared__ __attribute__((aligned(16))) uint4 Qs[GMAX * LPT];
    __shared__ float S[NG][GMAX][VOX_TC];
    __shared__ float Ms[NG][GMAX];
    __shared__ float Sh[16 * D];
    __shared__ __attribute__((aligned(16))) bf16_t Knew[D];
    __shared__ float Po[NCH][GMAX][D];
    __shared__ float2 Pml[NCH][GMAX];

    // head split: gridDim.x = Hkv * HS blocks per row, block (hk, hs) takes q heads hs * G .. hs * G + G - 1 of kv head hk's group
    const int HS = gridDim.x / a.Hkv, hk = blockIdx.x / HS, hs = blockIdx.x % HS, row = blockIdx.y;
    VOX_STAMP_DECL
    VOX_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63, wave16 = tid >> 6;
    const int grp = tid / GT, gt = tid % GT, gw = wave16 % GW;
    const int* pages = a.identity_pages ? nullptr
                       : (a.ptab ? a.ptab + (size_t)row * a.pt_stride : a.indices + a.indptr[a.q_req[row]]);
    // per-row page table: the page ids of this thread's tokens do not depend on the row's length, so they are requested
    // together with it (one exposed round trip in front of the K/V loads instead of two); entries past the row's last page are
    // read (clamped to the table row) and never used
    int pgi_pre[CPG][KVL];
    const bool hoist = a.ptab && a.hoist;
    if (hoist) {
#pragma unroll
        for (int ci = 0; ci < CPG; ++ci)
#pragma unroll
            for (int u = 0; u < KVL; ++u) {
                const int tok = (grp + NG * ci) * VOX_TC + (gt + GT * u) / LPT;
                const int pi = page_of(a, tok);
                pgi_pre[ci][u] = pages[pi < a.pt_stride ? pi : a.pt_stride - 1];
            }
    }
    const int L = a.fixed_kvlen > 0 ? a.fixed_kvlen : a.q_kvlen[row];
    // (the new token's page / slot words with the row length: one exposed round trip less in front of the append — attn_decode8_block)
    const int pg_new = a.identity_pages ? row : a.page[row];
    const int sl_new = a.identity_pages ? 0 : a.slot[row];
    const int nc = (L + VOX_TC - 1) / VOX_TC;                 // 1..8
    const int Gf = a.Hq / a.Hkv, G = Gf / HS, g0 = hs * G;
    const size_t ps = (size_t)2 * a.page_size * a.Hkv * D;

    // K/V tiles of this group's chunks (the row's newest token, index L - 1, comes from the projection output instead)
    uint4 kreg[CPG][KVL], vreg[CPG][KVL];
#pragma unroll
    for (int ci = 0; ci < CPG; ++ci) {
        const int t0 = (grp + NG * ci) * VOX_TC;
#pragma unroll
        for (int u = 0; u < KVL; ++u) {
            const int i = gt + GT * u, t = i / LPT, j = i % LPT;
            kreg[ci][u] = make_uint4(0, 0, 0, 0);
            vreg[ci][u] = kreg[ci][u];
            const int tok = t0 + t;
            if (i < VOX_TC * LPT && tok < L - 1) {
                const int pgi = hoist ? pgi_pre[ci][u] : (pages ? pages[page_of(a, tok)] : row);
                const bf16_t* base = a.kv + (size_t)pgi * ps + ((size_t)slot_of(a, tok) * a.Hkv + hk) * D;
                kreg[ci][u] = reinterpret_cast<const uint4*>(base)[j];
                vreg[ci][u] = reinterpret_cast<const uint4*>(base + (size_t)a.page_size * a.Hkv * D)[j];
            }
        }
    }
    VOX_STAMP(1)
    {   // q heads of this kv head (per-head norm + RoPE) and the new k: one head per wave
        const int nqkv = (a.Hq + 2 * a.Hkv) * D;
        const bf16_t* raw = a.qkv + (size_t)row * nqkv;
        int p = a.fixed_pos >= 0 ? a.fixed_pos : a.pos[row];
        p = p < 0 ? 0 : (p >= a.table_max_pos ? a.table_max_pos - 1 : p);
        const float* cs_row = a.cs ? a.cs + (size_t)p * (a.rot >> 1) * 2 : nullptr;
        for (int h = wave16; h < G + 1; h += 16) {
            const bool isk = h == G;
            const bf16_t* src = isk ? raw + (size_t)a.Hq * D + (size_t)hk * D : raw + (size_t)(hk * Gf + g0 + h) * D;
            bf16_t* dst = isk ? Knew : reinterpret_cast<bf16_t*>(Qs) + (size_t)h * D;
            prep_head<D>(src, isk ? a.kn : a.qn, a.eps, cs_row, a.rot, a.interleave, Sh + wave16 * D, dst, lane);
        }
    }
    VOX_STAMP(2)
#pragma unroll
    for (int ci = 0; ci < CPG; ++ci) {
        const int c = grp + NG * ci, t0 = c * VOX_TC;
        const bool live = t0 < L;
        const int nt = live ? ((L - t0) < VOX_TC ? (L - t0) : VOX_TC) : 0;
        const bool own_last = live && (t0 + nt == L);
        if (ci > 0) __syncthreads();           // the previous chunk's tile is dead
#pragma unroll
        for (int u = 0; u < KVL; ++u) {
            const int i = gt + GT * u;
            if (i < VOX_TC * LPT) { Ks[grp][i] = kreg[ci][u]; Vs[grp][i] = vreg[ci][u]; }
        }
        __syncthreads();                       // tiles parked; (ci = 0) Qs / Knew written
        VOX_STAMP(3)
        if (own_last) {
            // place the new token into the tile and append it to the paged cache (page < 0: graph padding row)
            const bf16_t* vraw = a.qkv + (size_t)row * (a.Hq + 2 * a.Hkv) * D + (size_t)(a.Hq + a.Hkv) * D + (size_t)hk * D;
            const int pg = pg_new;
            const int sl = a.identity_pages ? (L - 1) : sl_new;
            if (gt < LPT) {
                const uint4 kx = reinterpret_cast<const uint4*>(Knew)[gt];
                const uint4 vx = reinterpret_cast<const uint4*>(vraw)[gt];
                Ks[grp][(nt - 1) * LPT + gt] = kx;
                Vs[grp][(nt - 1) * LPT + gt] = vx;
                if (pg >= 0 && hs == 0) {
                    bf16_t* base = a.kv_w + (size_t)pg * ps + ((size_t)sl * a.Hkv + hk) * D;
                    reinterpret_cast<uint4*>(base)[gt] = kx;
                    reinterpret_cast<uint4*>(base + (size_t)a.page_size * a.Hkv * D)[gt] = vx;
                }
            }
        }
        __syncthreads();
        VOX_STAMP(4)
        if (live) {      // scores: LPT lanes per token, butterfly over LPT lanes
#pragma unroll
            for (int tb = gw * TPW; tb < VOX_TC; tb += GW * TPW) {
                const int tt = tb + lane / LPT, j = lane % LPT;
                const uint4 kx = Ks[grp][tt * LPT + j];
                for (int g = 0; g < G; ++g) {
                    float d = dot8(Qs[g * LPT + j], kx, 0.0f);
                    d = butterfly<LPT>(d);
                    if (j == 0) S[grp][g][tt] = d * a.scale;
                }
            }
        }
        __syncthreads();
        VOX_STAMP(5)
        if (live) {      // chunk max + p = exp2((s-m)*log2e): 32 lanes per q head
            for (int pr = gt; pr < G * VOX_TC; pr += GT) {
                const int g = pr / VOX_TC, t = pr % VOX_TC;
                const float s = t < nt ? S[grp][g][t] : -INFINITY;
                float m = s;
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, VOX_WAVE));
                const float p = t < nt ? exp2_c((s - m) * VOX_LOG2E) : 0.0f;
                S[grp][g][t] = p;
                if (t == 0) Ms[grp][g] = m;
            }
        }
        __syncthreads();
        VOX_STAMP(6)
        if (live) {      // PV: one thread per (q head, d); sequential over the tokens of the (zero-padded) tile
            const bf16_t* Vb = reinterpret_cast<const bf16_t*>(Vs[grp]);
            for (int e = gt; e < G * D; e += GT) {
                const int g = e / D, d = e % D;
                float o = 0.0f, l = 0.0f;
#pragma unroll
                for (int t = 0; t < VOX_TC; ++t) {
                    const float p = S[grp][g][t];          // 0 for t >= nt
                    l = l + p;
                    o = __fmaf_rn(p, bf2f(Vb[t * D + d]), o);
                }
                Po[c][g][d] = o;
                if (d == 0) Pml[c][g] = make_float2(Ms[grp][g], l);
            }
        }
    }
    __syncthreads();
    VOX_STAMP(7)
    // merge (k_attn_merge): global max, then L and O over the chunks in ascending order
    for (int e = tid; e < G * D; e += 1024) {
        const int g = e / D, d = e % D;
        float M = -INFINITY, Lsum = 0.0f, O = 0.0f;
        for (int c = 0; c < nc; ++c) M = fmaxf(M, Pml[c][g].x);
        for (int c = 0; c < nc; ++c) {
            const float w = exp2_c((Pml[c][g].x - M) * VOX_LOG2E);
            Lsum = __fmaf_rn(Pml[c][g].y, w, Lsum);
            O = __fmaf_rn(Po[c][g][d], w, O);
        }
        const bf16_t r = f2bf(O / Lsum);
        const int h = hk * Gf + g0 + g;
        a.out[((size_t)row * a.Hq + h) * D + d] = r;
        if (a.out_frag) a.out_frag[frag_off(row, h * D + d, a.Hq * D)] = r;
    }
    VOX_STAMP(8)
}

#endif

// rows from which contexts of 257..512 tokens take the one-launch form (below: partial + merge — at one row 8 blocks running two
// rounds lose to 88 chunk blocks: 2.79 vs 2.59 ms per frame; at 32 rows 4.09 vs 4.27).  VOX_ATTN_DECODE16 = 0 never, n = from n rows
// (read per call — launches are captured into graphs, this is not a hot path — so that a test can A/B it)
static int decode16_min_rows() {
    const char* e = getenv("VOX_ATTN_DECODE16");
    if (!e) return 16;      // measured at kv 330: 4 / 8 / 12 rows slower (+4 .. +2 %), 16 equal (round 6, one address computation per tile: -1.8 %), 24 -1.9 %, 32 -4.2 %
    const int v = atoi(e);
    return v <= 0 ? (1 << 30) : v;
}
// true when the one-launch decode attention covers the call (fused decode rows, <= 8 chunks, a supported head shape)
bool vox_attn_decode8_supported(const AttnCall& c) {
    if (!c.qkv || !c.out || c.Nq < 1 || c.Hkv < 1 || c.Hq % c.Hkv) return false;
    const int nchunk = (c.max_kvlen + VOX_TC - 1) / VOX_TC, G = c.Hq / c.Hkv;
    if (nchunk < 2 || nchunk > 16) return false;
    if (nchunk > 8) return c.Nq >= decode16_min_rows() && ((c.D == 128 && G == 2) || (c.D == 64 && G == 4));      // 257..512 tokens: two chunks per group
    return (c.D == 128 && (G == 2 || G == 16)) || (c.D == 64 && (G == 4 || G == 7));
}
int vox_launch_attn_decode8(hipStream_t st, const AttnCall& c) {
    if (!vox_attn_decode8_supported(c)) return vox_fail(VOX_ERR_INVALID, "attn_decode8: unsupported shape");
    AttnArgs a{};
    a.q = (const bf16_t*)c.q; a.kv = (const bf16_t*)c.kv; a.q_req = c.q_req; a.q_kvlen = c.q_kvlen;
    a.indptr = c.indptr; a.indices = c.indices; a.scale = c.scale;
    a.Hq = c.Hq; a.Hkv = c.Hkv; set_page_size(a, c.page_size); a.max_chunks = c.max_chunks;
    a.qkv = (const bf16_t*)c.qkv; a.kv_w = (bf16_t*)const_cast<void*>(c.kv); a.qn = (const bf16_t*)c.qn;
    a.kn = (const bf16_t*)c.kn; a.cs = c.cs; a.pos = c.pos; a.page = c.page; a.slot = c.slot; a.eps = c.eps;
    a.rot = c.rot; a.interleave = c.interleave; a.table_max_pos = c.table_max_pos;
    a.ptab = c.ptab; a.pt_stride = c.pt_stride; a.fixed_kvlen = c.fixed_kvlen; a.fixed_pos = c.fixed_pos;
    a.identity_pages = c.identity_pages;
    a.out = (bf16_t*)c.out; a.out_frag = (bf16_t*)c.out_frag;
    static const int hoist_on = [] { const char* e = getenv("VOX_ATTN_HOIST"); return !(e && e[0] == '0'); }();
    a.hoist = hoist_on;
#ifdef VOX_DEV_KNOBS
    static const int hnd = [] { const char* e = getenv("VOX_ATTN_HND_PROBE"); return e && e[0] == '1'; }();
    a.hnd_probe = hnd;
#endif
    const int G = c.Hq / c.Hkv;
    if ((c.max_kvlen + VOX_TC - 1) / VOX_TC > 8) {      // 257..512 visible tokens
        const dim3 grid16(c.Hkv, c.Nq);
        if (c.D == 128) hipLaunchKernelGGL((k_attn_decode8<128, 2, 8, 16>), grid16, dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((k_attn_decode8<64, 4, 8, 16>), grid16, dim3(1024), 0, st, a);
        return VOX_OK;
    }
    // 16-head groups (GLM-4-Voice: 2 kv heads, so 2 blocks per row): eight blocks of two q heads per kv head, each with all 8 chunks in
    // flight — the per-(row, head) arithmetic does not depend on which heads share a block (VOX_ATTN_HEADSPLIT=0: one block per kv head)
    static const bool split_on = [] { const char* e = getenv("VOX_ATTN_HEADSPLIT"); return !(e && e[0] == '0'); }();
    if (c.D == 128 && G == 16 && split_on) {
        hipLaunchKernelGGL((k_attn_decode8<128, 2, 8>), dim3(c.Hkv * 8, c.Nq), dim3(1024), 0, st, a);
        return VOX_OK;
    }
    // CosyVoice2's 7-head groups on 2 kv heads (2 blocks per row): one block per q head
    if (c.D == 64 && G == 7 && split_on) {
        hipLaunchKernelGGL((k_attn_decode8<64, 1, 8>), dim3(c.Hkv * 7, c.Nq), dim3(1024), 0, st, a);
        return VOX_OK;
    }
    // two-head groups (Qwen3-TTS talker, CSM, Orpheus) at few rows: one block per q head (Hq blocks per row instead of Hkv)
    static const int hs2_rows = [] { const char* e = getenv("VOX_ATTN_HS2_ROWS"); return e ? atoi(e) : 4; }();      // measured: -1.2 us per talker layer at 1 row, no gain from 8 rows on, slower at 32
    if (c.D == 128 && G == 2 && c.Nq <= hs2_rows) {
        hipLaunchKernelGGL((k_attn_decode8<128, 1, 8>), dim3(c.Hkv * 2, c.Nq), dim3(1024), 0, st, a);
        return VOX_OK;
    }
    const dim3 grid(c.Hkv, c.Nq);
#ifdef VOX_DEV_KNOBS
    static const bool v1 = [] { const char* e = getenv("VOX_ATTN_V1"); return e && e[0] == '1'; }();
    if (v1 && c.D == 128 && G == 2) { hipLaunchKernelGGL((k_attn_decode8_v1<128, 2, 8>), grid, dim3(1024), 0, st, a); return VOX_OK; }
#endif
#define VOX_AD(D_, G_, NG_) if (c.D == D_ && G == G_) { hipLaunchKernelGGL((k_attn_decode8<D_, G_, NG_>), grid, dim3(1024), 0, st, a); return VOX_OK; }
    VOX_AD(128, 2, 8) VOX_AD(128, 16, 4) VOX_AD(64, 4, 8) VOX_AD(64, 7, 8)
#undef VOX_AD
    return vox_fail(VOX_ERR_INVALID, "attn_decode8: no variant");
}

// ================================================================================================
// Short-context decode attention (depth transformer: <= 16 visible tokens, D = 128, two q heads per kv head).
// One WAVE per (row, kv head), no LDS, no barrier.  Lane = 16*grp + j holds 16-byte chunk j of: q head 0 (grp 0),
// q head 1 (grp 1), the new k (grp 2), the new v (grp 3) — norm, RoPE (partner chunk = lane ^ 8) and the dot products
// all stay in registers; K sits token-major (token 4u+grp in pass u), V chunk-major.  Same arithmetic and order as
// k_attn_partial with one chunk (canonical DOT over 16 lanes, sequential l / o over tokens), bit for bit.
__device__ __forceinline__ uint4 shfl4(uint4 v, int src) {
    return make_uint4(__shfl(v.x, src, VOX_WAVE), __shfl(v.y, src, VOX_WAVE), __shfl(v.z, src, VOX_WAVE),
                      __shfl(v.w, src, VOX_WAVE));
}
// NT > 0: the number of visible tokens is a compile-time constant (depth loop: step i sees exactly i + 1 tokens, known when
// the frame graph is captured): every token loop has its exact trip count and no predicates — at one wave per (row, head)
// instruction count is time.  NT == 0: read from the plan arrays (<= 16).
// Operands of the short attention that do NOT depend on the row's fresh q / k / v: the cached K (token-major: token 4u + grp, 16-byte
// chunk j) and V (P.V layout: 4 dims per lane) of the earlier tokens, the norm weight chunk and the RoPE table entries.  Requested
// first, so that they are in flight while the projection output arrives (plain loads in the launch-per-stage kernels, a polled
// hand-off in the persistent depth step).
template <int NT>
struct AttnShortPre {
    static constexpr int TMAX = NT > 0 ? NT : 16, UMAX = (TMAX + 3) / 4;
    uint4 kr[UMAX];
    uint2 vr[TMAX];
    uint4 gw4;
    float4 cs4[4];
    int L, nt;
};
// PART (fast path only): 0 = every operand, 1 = all but the cached V rows, 2 = the cached V rows alone.  The persistent depth step asks for
// the V rows BEHIND its qkv gather: they are needed a microsecond later than K (after scores + softmax), and in front of the gather their
// 15 requests per wave stood between the block and its polls (loads return in order: 0.1 us per visible token on every hand-off).
template <int NT, bool FAST = true, int PART = 0>
__device__ __forceinline__ void attn_short_prefetch(const AttnArgs& at, int row, int hk, int lane, AttnShortPre<NT>& pf) {
    constexpr int D = 128, TMAX = AttnShortPre<NT>::TMAX, UMAX = AttnShortPre<NT>::UMAX;
    const int grp = lane >> 4, j = lane & 15, dq = lane & 31;
    const size_t ps = (size_t)2 * at.page_size * at.Hkv * D;
    if constexpr (NT > 0 && FAST && PART == 2) {
        const bf16_t* vb = at.kv + (size_t)row * ps + (size_t)hk * D + (size_t)at.page_size * at.Hkv * D;
        const size_t ts = (size_t)at.Hkv * D;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            pf.vr[t] = make_uint2(0, 0);
            if (t < NT - 1) pf.vr[t] = reinterpret_cast<const uint2*>(vb + (size_t)t * ts)[dq];
        }
        return;
    }
    if constexpr (NT > 0 && FAST) {
        // Depth loop (the launchers pick NT > 0 only with identity pages, a fixed position and page_size >= NT): page = row, slot = token,
        // position = fixed_pos — no plan array is read, every address is base + constant * stride, and every request of the wave goes out
        // back to back.  (Through the general path below each token cost a page-table select, a shift / divide select and a conditional
        // request with a wait of its own: ~95 instructions per token in front of a 300-instruction kernel — at one wave per (row, head)
        // instruction count is time.)  A lane whose token is not visible reads the LAST cached token instead of nothing (no predicate on
        // the request); its score is masked to -inf and its K never used, as before.
        pf.L = NT; pf.nt = NT;
        const bf16_t* nwp = grp < 2 ? at.qn : at.kn;                    // (grp 3 = the new v takes no norm: its weight chunk is not used)
        pf.gw4 = make_uint4(0, 0, 0, 0);
        if (at.qn && at.kn) pf.gw4 = ldg(reinterpret_cast<const uint4*>(nwp) + j);      // uniform condition
        int p = at.fixed_pos;
        p = p < 0 ? 0 : (p >= at.table_max_pos ? at.table_max_pos - 1 : p);
        const float4* cp = reinterpret_cast<const float4*>(at.cs + (size_t)p * D) + (j & 7) * 4;   // row = D/2 (c,s) pairs
#pragma unroll
        for (int k = 0; k < 4; ++k) pf.cs4[k] = cp[k];
        const size_t ts = (size_t)at.Hkv * D;                            // elements between two tokens of a page
        const bf16_t* kb = at.kv + (size_t)row * ps + (size_t)hk * D;
        const bf16_t* vb = kb + (size_t)at.page_size * at.Hkv * D;
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            int t = u * 4 + grp;
            if (u * 4 + 3 >= NT - 1) t = t < NT - 1 ? t : (NT >= 2 ? NT - 2 : 0);      // only the last pass can run past the cached tokens
            pf.kr[u] = NT >= 2 ? reinterpret_cast<const uint4*>(kb + (size_t)t * ts)[j] : make_uint4(0, 0, 0, 0);
        }
        if constexpr (PART == 0) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                pf.vr[t] = make_uint2(0, 0);
                if (t < NT - 1) pf.vr[t] = reinterpret_cast<const uint2*>(vb + (size_t)t * ts)[dq];
            }
        }
        return;
    }
    pf.L = NT > 0 ? NT : (at.fixed_kvlen > 0 ? at.fixed_kvlen : at.q_kvlen[row]);
    pf.nt = NT > 0 ? NT : (pf.L < TMAX ? pf.L : TMAX);
    const int nt = pf.nt;
    const int* pages = at.identity_pages ? nullptr
                       : (at.ptab ? at.ptab + (size_t)row * at.pt_stride : at.indices + at.indptr[at.q_req[row]]);
    const bf16_t* nwp = grp < 2 ? at.qn : (grp == 2 ? at.kn : nullptr);
    pf.gw4 = make_uint4(0, 0, 0, 0);
    if (nwp) pf.gw4 = reinterpret_cast<const uint4*>(nwp)[j];
    int p = at.fixed_pos >= 0 ? at.fixed_pos : at.pos[row];
    p = p < 0 ? 0 : (p >= at.table_max_pos ? at.table_max_pos - 1 : p);
    {
        const float4* cp = reinterpret_cast<const float4*>(at.cs + (size_t)p * D) + (j & 7) * 4;   // row = D/2 (c,s) pairs
#pragma unroll
        for (int k = 0; k < 4; ++k) pf.cs4[k] = cp[k];
    }
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
        const int t = u * 4 + grp;
        pf.kr[u] = make_uint4(0, 0, 0, 0);
        if (t < nt - 1) {
            const int pgi = pages ? pages[page_of(at, t)] : row;
            pf.kr[u] = reinterpret_cast<const uint4*>(at.kv + (size_t)pgi * ps + ((size_t)slot_of(at, t) * at.Hkv + hk) * D)[j];
        }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        pf.vr[t] = make_uint2(0, 0);
        if (t < nt - 1) {
            const int pgi = pages ? pages[page_of(at, t)] : row;
            pf.vr[t] = reinterpret_cast<const uint2*>(at.kv + (size_t)pgi * ps + (size_t)at.page_size * at.Hkv * D +
                                                      ((size_t)slot_of(at, t) * at.Hkv + hk) * D)[dq];
        }
    }
}

// v: this lane's 16-byte chunk of its group's head row (grp 0 / 1: the two q heads, 2: the new k, 3: the new v);
// vnew: the new token's V row in the P.V layout (dims 4 dq .. 4 dq + 3).
template <int NT>
__device__ __forceinline__ void attn_short_compute(const AttnArgs& at, int row, int hk, int lane, const uint4 v, const uint2 vnew,
                                                   const AttnShortPre<NT>& pf, bf16_t* out_row, bool do_append, int hq0 = -1) {
    if (hq0 < 0) hq0 = hk * 2;           // the wave's two q heads: hq0, hq0 + 1 (two q heads per kv head: 2 hk; four: 4 hk + 2 pair)
    constexpr int D = 128, TMAX = AttnShortPre<NT>::TMAX, UMAX = AttnShortPre<NT>::UMAX;
    const int grp = lane >> 4, j = lane & 15;
    const int g = lane >> 5, dq = lane & 31;      // P.V / output layout: lane = (q head g, dims 4 dq .. 4 dq + 3)
    const size_t ps = (size_t)2 * at.page_size * at.Hkv * D;
    const int L = pf.L, nt = pf.nt;
    // (two uniform conditions selected by lane group: a lane-indexed select of the two POINTERS goes through a scratch array, and
    // the wait behind that scratch load waits for every K/V request in flight)
    const bool qn_ok = at.qn != nullptr, kn_ok = at.kn != nullptr;
    const bool has_nw = grp < 2 ? qn_ok : (grp == 2 && kn_ok);
    const uint4 gw4 = pf.gw4;
    // per-head RMSNorm (prep_head: butterfly<64> over a head's 16 non-zero lanes == butterfly<16>)
    float e[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
    {
        float s = sq8(v, 0.0f);
        s = butterfly<16>(s);
        const float rinv = 1.0f / sqrtf(s / (float)D + at.eps);
        if (has_nw) {
            const float gw[8] = {bflo(gw4.x), bfhi(gw4.x), bflo(gw4.y), bfhi(gw4.y), bflo(gw4.z), bfhi(gw4.z), bflo(gw4.w), bfhi(gw4.w)};
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = bfround((e[i] * rinv) * gw[i]);
        }
    }
    // NeoX RoPE over the full head: element i < 64 pairs with i + 64, i.e. with the same slot of lane ^ 8 (a rotation by 8 inside
    // the 16-lane row: one DPP move)
    uint4 hq;
    {
        const float cc[8] = {pf.cs4[0].x, pf.cs4[0].z, pf.cs4[1].x, pf.cs4[1].z, pf.cs4[2].x, pf.cs4[2].z, pf.cs4[3].x, pf.cs4[3].z};
        const float sn[8] = {pf.cs4[0].y, pf.cs4[0].w, pf.cs4[1].y, pf.cs4[1].w, pf.cs4[2].y, pf.cs4[2].w, pf.cs4[3].y, pf.cs4[3].w};
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#if VOX_DPP_BUTTERFLY
            const float o = vox_dpp<0x128, 0xF>(e[i], e[i]);        // row_ror:8
#else
            const float o = __shfl_xor(e[i], 8, VOX_WAVE);
#endif
            const float mc = e[i] * cc[i];
            r[i] = (j < 8) ? __fmaf_rn(-o, sn[i], mc) : __fmaf_rn(o, sn[i], mc);
        }
        hq.x = pack_bf2(r[0], r[1]); hq.y = pack_bf2(r[2], r[3]); hq.z = pack_bf2(r[4], r[5]); hq.w = pack_bf2(r[6], r[7]);
        if (grp == 3) hq = v;
    }
    const uint4 q0c = shfl4(hq, j), q1c = shfl4(hq, 16 + j), knc = shfl4(hq, 32 + j);
    if (do_append) {
        const int pg = (NT > 0 || at.identity_pages) ? row : at.page[row];
        const int sl = (NT > 0 || at.identity_pages) ? (L - 1) : at.slot[row];
        if (pg >= 0 && grp >= 2) {
            bf16_t* base = at.kv_w + (size_t)pg * ps + ((size_t)sl * at.Hkv + hk) * D;
            if (grp == 3) base += (size_t)at.page_size * at.Hkv * D;
            reinterpret_cast<uint4*>(base)[j] = hq;
        }
    }
    // scores of token 4u+grp for both q heads (the q chunks unpacked once, the token's K chunk once per pass), then the global max
    const float q0f[8] = {bflo(q0c.x), bfhi(q0c.x), bflo(q0c.y), bfhi(q0c.y), bflo(q0c.z), bfhi(q0c.z), bflo(q0c.w), bfhi(q0c.w)};
    const float q1f[8] = {bflo(q1c.x), bfhi(q1c.x), bflo(q1c.y), bfhi(q1c.y), bflo(q1c.z), bfhi(q1c.z), bflo(q1c.w), bfhi(q1c.w)};
    float sc[2][UMAX], m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
        const int t = u * 4 + grp;
        const uint4 kx = (t == nt - 1) ? knc : pf.kr[u];
        const float kf[8] = {bflo(kx.x), bfhi(kx.x), bflo(kx.y), bfhi(kx.y), bflo(kx.z), bfhi(kx.z), bflo(kx.w), bfhi(kx.w)};
        float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { d0 = __fmaf_rn(q0f[i], kf[i], d0); d1 = __fmaf_rn(q1f[i], kf[i], d1); }
        d0 = butterfly<16>(d0) * at.scale;
        d1 = butterfly<16>(d1) * at.scale;
        sc[0][u] = t < nt ? d0 : -INFINITY;
        sc[1][u] = t < nt ? d1 : -INFINITY;
        m0 = fmaxf(m0, sc[0][u]);
        m1 = fmaxf(m1, sc[1][u]);
    }
    m0 = fmaxf(m0, __shfl_xor(m0, 16, VOX_WAVE)); m0 = fmaxf(m0, __shfl_xor(m0, 32, VOX_WAVE));
    m1 = fmaxf(m1, __shfl_xor(m1, 16, VOX_WAVE)); m1 = fmaxf(m1, __shfl_xor(m1, 32, VOX_WAVE));
    // p = exp2((s - m) log2e): the scores of a token are replicated over the 16 lanes of its group, so lane j < 2 UMAX of group
    // grp evaluates ONE of them — head j / UMAX, token 4 (j % UMAX) + grp — instead of every lane all 2 UMAX
    float pe;
    {
        float ssel = -INFINITY;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < UMAX; ++u) ssel = (j == h * UMAX + u) ? sc[h][u] : ssel;
        const float msel = j >= UMAX ? m1 : m0;
        const int tl = (j >= UMAX ? j - UMAX : j) * 4 + grp;
        pe = (j < 2 * UMAX && tl < nt) ? exp2_c((ssel - msel) * VOX_LOG2E) : 0.0f;
    }
    // P.V: lane (g, dq) accumulates its 4 output dims sequentially over the tokens
    float o[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.0f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        // p of (token t, head h) sits in lane 16 (t & 3) + h UMAX + (t >> 2): fetched with v_readlane (a scalar broadcast)
        const float p0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pe), (t & 3) * 16 + (t >> 2)));
        const float p1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pe), (t & 3) * 16 + UMAX + (t >> 2)));
        if (t < nt) {
            const float pt = g ? p1 : p0;
            const uint2 vx = (t == nt - 1) ? vnew : pf.vr[t];
            l = l + pt;
            o[0] = __fmaf_rn(pt, bflo(vx.x), o[0]); o[1] = __fmaf_rn(pt, bfhi(vx.x), o[1]);
            o[2] = __fmaf_rn(pt, bflo(vx.y), o[2]); o[3] = __fmaf_rn(pt, bfhi(vx.y), o[3]);
        }
    }
    {
        uint2 r;
        r.x = pack_bf2(o[0] / l, o[1] / l); r.y = pack_bf2(o[2] / l, o[3] / l);
        const int col = (hq0 + g) * D + 4 * dq;
        *reinterpret_cast<uint2*>(out_row + col) = r;
        if (at.out_frag) *reinterpret_cast<uint2*>(at.out_frag + frag_off(row, col, at.Hq * D)) = r;
    }
}


// launch-per-stage form: the row's q / k / v come from the projection output in memory
template <int NT>
__device__ __forceinline__ void attn_short_wave(const AttnArgs& at, int row, int hk, int lane, bf16_t* out_row,
                                                bool do_append, int hq0 = -1) {
    if (hq0 < 0) hq0 = hk * 2;
    constexpr int D = 128;
    const int grp = lane >> 4, j = lane & 15, dq = lane & 31;
    const int nqkv = (at.Hq + 2 * at.Hkv) * D;
    const bf16_t* raw = at.qkv + (size_t)row * nqkv;
    const bf16_t* src = grp == 0 ? raw + (size_t)hq0 * D
                      : grp == 1 ? raw + (size_t)(hq0 + 1) * D
                      : grp == 2 ? raw + (size_t)at.Hq * D + (size_t)hk * D
                                 : raw + (size_t)(at.Hq + at.Hkv) * D + (size_t)hk * D;
    const uint4 v = reinterpret_cast<const uint4*>(src)[j];
    // the new token's V row in the P.V layout (4 dims per lane), straight from the projection output
    const uint2 vnew = reinterpret_cast<const uint2*>(raw + (size_t)(at.Hq + at.Hkv) * D + (size_t)hk * D)[dq];
    AttnShortPre<NT> pf;
    attn_short_prefetch<NT>(at, row, hk, lane, pf);
    attn_short_compute<NT>(at, row, hk, lane, v, vnew, pf, out_row, do_append, hq0);
}

// standalone: one wave per (row, kv head, PAIR of its q heads), four waves per block (NT as in attn_short_wave).  Two q heads per kv head
// (Qwen3-TTS depth decoder): one pair.  Four (CSM-1B depth decoder, up to 32 visible tokens): two waves per (row, kv head) — each reads the
// head's cached K / V itself (L2 hits), the first one appends the new token.
template <int NT>
__global__ __launch_bounds__(256) void k_attn_short(AttnArgs at, int n_pairs) {
    VOX_TR_DECL
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= n_pairs) return;
    const int PP = at.Hq / (2 * at.Hkv), per_row = at.Hkv * PP;
    const int row = pi / per_row, rem = pi - row * per_row, hk = rem / PP, pp = rem - hk * PP;
    attn_short_wave<NT>(at, row, hk, threadIdx.x & 63, at.out + (size_t)row * at.Hq * 128, pp == 0, hk * 2 * PP + 2 * pp);
    VOX_TR_END(2, NT)
}

// Fused into the o_proj GEMV: every o_proj block recomputes the row's attention (8 waves: one (row, kv head) pair
// each per round) into LDS while its weight rows are in flight, block 0 appends the new K/V to the cache.
template <int BT, int KC, int R, int NT>
__global__ __launch_bounds__(512) void k_attn1_linear(AttnArgs at, LinArgs a) {
    __shared__ __attribute__((aligned(16))) uint4 xs[BT * KC * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 8 + wave) * R;
    uint4 w[R][KC];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int n = n0 + r;
        n = n < a.N ? n : a.N - 1;
        const uint4* wr = reinterpret_cast<const uint4*>(a.W + (size_t)n * a.K);
#pragma unroll
        for (int j = 0; j < KC; ++j) w[r][j] = wr[lane + 64 * j];     // depth weights: re-read every step, keep cached
    }
    float res_pre = 0.0f, bias_pre = 0.0f;
    {
        const int po = lane / BT, pb = lane % BT;
        if (lane < R * BT && pb < a.B && n0 + po < a.N) {
            if (a.residual) res_pre = bf2f(a.residual[(size_t)pb * a.N + n0 + po]);
            if (a.bias) bias_pre = bf2f(a.bias[n0 + po]);
        }
    }
    for (int pi = wave; pi < a.B * at.Hkv; pi += 8) {
        const int row = pi / at.Hkv, hk = pi % at.Hkv;
        attn_short_wave<NT>(at, row, hk, lane, reinterpret_cast<bf16_t*>(xs) + (size_t)row * a.K, blockIdx.x == 0);
    }
    __syncthreads();
    if (n0 >= a.N) return;
    float acc[R][BT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            float sacc = 0.0f;
#pragma unroll
            for (int j = 0; j < KC; ++j) sacc = dot8(w[r][j], xs[b * KC * 64 + lane + 64 * j], sacc);
            acc[r][b] = butterfly<64>(sacc);
        }
#pragma unroll
    for (int o = 0; o < R; ++o)
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            if (lane == o * BT + b && b < a.B && n0 + o < a.N) {
                const int n = n0 + o;
                float v = acc[o][b];
                if (a.bias) v = v + bias_pre;
                bf16_t r = f2bf(v);
                if (a.residual) r = f2bf(res_pre + bf2f(r));
                a.y[(size_t)b * a.N + n] = r;
            }
        }
}

// ================================================================================================
// Persistent depth step (one request): the 5 layers x 4 stages + the codebook head of ONE depth-loop step in a single launch of
// 256 resident blocks, instead of 21 dependent launches.  A stage's output vector travels between the blocks as 8-byte granules
// {2 x bf16, tag} written by one relaxed agent-scope store each (sc1 write-through) and polled with relaxed agent-scope loads
// until every tag matches (MI355X guide, persistent-kernel price list: granule hand-off); the weights of a stage are requested
// BEFORE its inputs are polled, so they stream while the previous stage finishes — that, and 20 kernel boundaries less per step,
// is the gain.  Arithmetic: k_gemv<1, KC, ...> / k_attn1_linear<1, 4, 1, NT> stage for stage — the same lane -> chunk assignment,
// the same sequential dot8 / butterfly<64> order, the same bf16 rounding points — so the step is bit-identical to the launch chain.
//   stage A  qkv    = Wqkv . rmsnorm(x, ln1)                         2048 column pairs: one per wave of every block
//   stage B  x     += Wo . attention(qkv, depth KV)                  attention recomputed by every block (8 waves = 8 kv heads)
//   stage C  h      = silu(Wg . n) * (Wu . n),  n = rmsnorm(x, ln2)  1536 pairs: waves 0..5
//   stage D  x     += Wd . h                                         512 pairs: waves 0, 1
//   head     logits = Whead . rmsnorm(x, norm)                       1024 pairs: waves 0..3, plain stores (a kernel boundary follows)
// Tags: epoch * 64 + stage id; the epoch word is read by every block when it starts and advanced by block 0 when it ends (no block
// can still be starting then: every block publishes a piece of stage A of layer 0, which block 0's later stages waited for).
// Every spin is bounded: on a timeout the error word is set and the block goes on (wrong data, never a hang).
struct DepthLayerW { const bf16_t *wqkv, *wo, *wgate, *wup, *wdown, *ln1, *ln2, *qn, *kn; };
struct DepthStepArgs {
    const DepthLayerW* layers;
    int n_layers;
    const bf16_t *final_norm, *head_w;
    const bf16_t* x_in;          // [1024] plain bf16: the step's input row
    bf16_t* logits;              // [2048] plain bf16
    unsigned long long *gx, *gqkv, *gh;   // granules: 512, 2048, 1536
    unsigned* epoch;
    unsigned* err;
    AttnArgs at;                 // kv / kv_w = layer 0 of the depth KV cache; fixed_pos, identity_pages, cs, eps, scale ... as in the launch chain
    long kv_layer_stride;        // elements between the layers' caches
    float eps;
    // the previous step's greedy pick at the start of this launch (DepthStepCall: pick_*)
    const bf16_t *pick_logits, *pick_tab, *pick_emb;
    int* pick_out;
    bf16_t* pick_feat;
    int pick_vocab, pick_H, pick_init;
    // the sampled form of the pick (pick_top_k > 0): DepthStepCall
    int pick_top_k;
    float pick_top_p, pick_min_p, pick_temperature;
    uint64_t pick_seed, pick_offset, pick_offset_mul;
    const uint64_t* pick_offset_dev;
    unsigned poll_delay;         // first-pass hold-back of the four gathers, one byte each (x 128 clocks): [7:0] x (D -> A / head), [15:8] qkv, [23:16] x (B -> C), [31:24] h
};
#define VOX_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// Bound of every poll loop (a pass is a round trip to the memory side, >= ~0.5 us: >= 20 ms).  Legitimate waits are microseconds —
// up to a fraction of a millisecond when blocks of the launch wait for CUs held by another stream's kernel.  On a timeout the error
// word is set and every later poll of the launch (and of later launches) gives up after one pass: wrong data, never a hang.
#ifndef VOX_PERSIST_SPINS
#define VOX_PERSIST_SPINS 40000u
#endif
// Gather of a granule vector into LDS: the block's last VOX_DS_PW waves poll (thread p of them takes granules p, p + 64 PW, ...) until
// the tags match and park the payload words; the caller's barrier publishes them.  Who polls, and how often, is part of the price:
// every poll is a round trip to the memory side, and pollers compete with the producers' weight streams and publishes (guide:
// polling-cost) — idle waves must wait at a barrier, not in a poll loop.
#ifndef VOX_DS_PW
#define VOX_DS_PW 8
#endif
#ifndef VOX_DS_SLEEP
#define VOX_DS_SLEEP 0
#endif
#ifndef VOX_DS_EXTRA_BARRIERS
#define VOX_DS_EXTRA_BARRIERS 1
#endif
// One poll pass over PER granules of a thread: the requests go out back to back and ONE wait follows.  Written as asm: as atomic loads the
// compiler interleaves each tag compare (and its wait) with the next request, and a predicated request gets a block of its own — a pass of
// four granules was three dependent round trips to the memory side, on the critical path of every hand-off.  Same instruction as
// __hip_atomic_load(relaxed, agent scope): global_load_dwordx2 ... sc1; payload and tag share the 8 bytes, so no ordering is needed.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int PER, int Q0 = 0, int Q1 = PER>
__device__ __forceinline__ void gran_poll_pass(const unsigned long long* const (&ptr)[PER], u32x2_t (&v)[PER]) {
    static_assert(PER >= 1 && PER <= 8, "gran_poll_pass lists the payload registers");
#pragma unroll
    for (int q = Q0; q < Q1; ++q) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[q]) : "v"(ptr[q]) : "memory");
    constexpr int N = Q1 - Q0;
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[Q0]) :: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[Q0]), "+v"(v[Q0 + 1]) :: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[Q0]), "+v"(v[Q0 + 1]), "+v"(v[Q0 + 2]) :: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[Q0]), "+v"(v[Q0 + 1]), "+v"(v[Q0 + 2]), "+v"(v[Q0 + 3]) :: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[Q0]), "+v"(v[Q0 + 1]), "+v"(v[Q0 + 2]), "+v"(v[Q0 + 3]), "+v"(v[Q0 + 4]) :: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[Q0]), "+v"(v[Q0 + 1]), "+v"(v[Q0 + 2]), "+v"(v[Q0 + 3]), "+v"(v[Q0 + 4]), "+v"(v[Q0 + 5]) :: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[Q0]), "+v"(v[Q0 + 1]), "+v"(v[Q0 + 2]), "+v"(v[Q0 + 3]), "+v"(v[Q0 + 4]), "+v"(v[Q0 + 5]), "+v"(v[Q0 + 6]) :: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[Q0]), "+v"(v[Q0 + 1]), "+v"(v[Q0 + 2]), "+v"(v[Q0 + 3]), "+v"(v[Q0 + 4]), "+v"(v[Q0 + 5]), "+v"(v[Q0 + 6]), "+v"(v[Q0 + 7]) :: "memory");
}
// The gather of a thread's PER granules.  VOX_GRAN_ASM = 2: the thread's FIRST granule is the sentinel — polled alone until its tag matches
// (a waiting block asks for 1 / PER of the vector per pass: polls travel the same fabric as the producers' weight streams and publishes),
// then the others in one batch (repeated for stragglers).  = 1: all PER per pass.
template <int PER, int TOTAL, typename InRange>
__device__ __forceinline__ void gran_poll_all(const unsigned long long* g, int tid, const unsigned long long* const (&gp)[PER], u32x2_t (&gv)[PER], unsigned tag,
                                              InRange in_range, unsigned* err, unsigned code, unsigned max_spins, unsigned delay = 0) {
    unsigned spin = 0;
    // The first pass is held back by `delay` x 128 clocks (uniform count: a scalar loop).  A hand-off takes ~2.5 us from the moment a block
    // is done with its own stage; every pass before the data can be there is traffic of 512 threads x 256 blocks on the fabric the producers
    // publish (and stream weights) through — polling EARLIER makes the hand-off LATER (round 6: a depth step whose K/V prefetch got ~1 us
    // shorter, so that its qkv gather started ~1 us sooner, ran 3 % slower; holding every gather's first pass back ~0.2-0.4 us: -3 %).
    for (unsigned i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(2);
    auto give_up = [&]() {
        if (spin > max_spins) { atomicCAS(err, 0u, code); return true; }
        if ((spin & 63u) == 63u && __hip_atomic_load(err, VOX_RLX_AGENT) != 0u) return true;      // somebody already gave up
        ++spin;
        if (VOX_DS_SLEEP) __builtin_amdgcn_s_sleep(VOX_DS_SLEEP);
        return false;
    };
    if constexpr (VOX_GRAN_ASM == 3) {
        // one wave polls 64 sentinels spread over the vector (every 1 / 64 of it: different producers), the others wait at the barrier
        if ((tid >> 6) == 7) {
            const unsigned long long* sp[1] = {g + (size_t)(tid & 63) * (TOTAL / 64)};
            u32x2_t sv[1];
            for (;;) {
                gran_poll_pass<1>(sp, sv);
                if (sv[0].y == tag || give_up()) break;
            }
        }
        __syncthreads();
        for (;;) {
            gran_poll_pass<PER>(gp, gv);
            bool ok = true;
#pragma unroll
            for (int q = 0; q < PER; ++q) ok = ok && (!in_range(q) || gv[q].y == tag);
            if (ok || give_up()) break;
        }
    } else if constexpr (VOX_GRAN_ASM == 4 && PER > 1) {
        // optimistic: behind a hold-back tuned to the hand-off's usual length the vector is normally THERE — ask for all PER granules at
        // once (one round trip instead of sentinel + batch = two); only when that pass misses fall back to the sentinel form
        for (;;) {
            gran_poll_pass<PER>(gp, gv);
            bool ok = true;
#pragma unroll
            for (int q = 0; q < PER; ++q) ok = ok && (!in_range(q) || gv[q].y == tag);
            if (ok || give_up()) break;
            for (;;) {
                gran_poll_pass<PER, 0, 1>(gp, gv);
                if (gv[0].y == tag || give_up()) break;
            }
        }
    } else if constexpr (VOX_GRAN_ASM == 2 && PER > 1) {
        for (;;) {
            gran_poll_pass<PER, 0, 1>(gp, gv);
            if (gv[0].y == tag || give_up()) break;
        }
        for (;;) {
            gran_poll_pass<PER, 1, PER>(gp, gv);
            bool ok = true;
#pragma unroll
            for (int q = 1; q < PER; ++q) ok = ok && (!in_range(q) || gv[q].y == tag);
            if (ok || give_up()) break;
        }
    } else {
        for (;;) {
            gran_poll_pass<PER>(gp, gv);
            bool ok = true;
#pragma unroll
            for (int q = 0; q < PER; ++q) ok = ok && (!in_range(q) || gv[q].y == tag);
            if (ok || give_up()) break;
        }
    }
}
template <int TOTAL>
__device__ __forceinline__ void gran_gather_lds(const unsigned long long* g, unsigned tag, unsigned* dst, int tid, unsigned* err, unsigned code, unsigned max_spins, unsigned delay = 0) {
    constexpr int NP = 64 * VOX_DS_PW, PER = (TOTAL + NP - 1) / NP;
    const int p = tid - (512 - NP);
    if (p < 0) return;
#if VOX_GRAN_ASM
    u32x2_t gv[PER];
    const unsigned long long* gp[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) gp[q] = g + (p + NP * q < TOTAL ? p + NP * q : p);      // (past the vector: the thread's first granule again, ignored)
    gran_poll_all<PER, TOTAL>(g, tid, gp, gv, tag, [&](int q) { return p + NP * q < TOTAL; }, err, code, max_spins, delay);
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (p + NP * q < TOTAL) dst[p + NP * q] = gv[q].x;
#else
    unsigned val[PER];
    for (unsigned spin = 0;; ++spin) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (p + NP * q < TOTAL) {
                const unsigned long long x = __hip_atomic_load(g + p + NP * q, VOX_RLX_AGENT);
                ok = ok && (unsigned)(x >> 32) == tag;
                val[q] = (unsigned)x;
            }
        }
        if (ok) break;
        if (spin > max_spins) { atomicCAS(err, 0u, code); break; }
        if ((spin & 63u) == 63u && __hip_atomic_load(err, VOX_RLX_AGENT) != 0u) break;      // somebody already gave up
        if (VOX_DS_SLEEP) __builtin_amdgcn_s_sleep(VOX_DS_SLEEP);
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (p + NP * q < TOTAL) dst[p + NP * q] = val[q];
#endif
}
__device__ __forceinline__ void gran_write(unsigned long long* g, unsigned tag, bf16_t lo, bf16_t hi) {
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned)lo | ((unsigned)hi << 16), VOX_RLX_AGENT);
}

template <int NT>
__global__ __launch_bounds__(512) void k_depth_step(DepthStepArgs a) {
    constexpr int H = 1024, NQKV = 4096, NQ = 2048, F = 3072;
    __shared__ __attribute__((aligned(16))) uint4 xb[H / 8];           // the residual stream x at the layer's input (bf16 row)
    __shared__ __attribute__((aligned(16))) uint4 xc[H / 8];           // ... after the attention half (two buffers: two barriers less per layer)
    __shared__ __attribute__((aligned(16))) uint4 qb[NQKV / 8];        // q | k | v of the step's token
    __shared__ __attribute__((aligned(16))) uint4 hb[F / 8];           // the FFN's activated row
    __shared__ __attribute__((aligned(16))) uint4 xs[NQ / 8];          // the attention output row
    // o_proj's weight rows of waves 0, 1, requested with stage A's and parked here when stage A is done: held in registers across the
    // attention (32 of them, next to its cached K / V rows) the step spilled from 7 visible tokens on — 124 bytes of scratch per lane at 16
    __shared__ __attribute__((aligned(16))) uint4 wos[2][2][4][64];
    // (wave index made provably uniform: the weight rows' base addresses then live in scalar registers — left as a per-lane value
    // the compiler hoists some forty 64-bit row pointers out of the layer loop into vector registers and spills them)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), blk = blockIdx.x;
    // The launch's state words beside the epoch — [2] the bound of every poll loop (VOX_PERSIST_SPINS unless the host lowered it for a
    // test), [3] the test hook "block 1 withholds its first publish of this launch" (vox_qwen3_persist_inject) — are read by ONE thread
    // per block (L1-bypassing loads: 2048 waves asking for the same line would queue) and reach the others through LDS at the first
    // barrier of stage A, in front of the launch's first publish.
    unsigned ep = 0, max_spins = 0, tag0 = 0;
    bool drop_first = false;
    auto tagof = [&](int l, int st) { return tag0 + 1u + (unsigned)(l * 4 + st); };
    VOX_STAMP2_DECL
    VOX_STAMP2(0)
    __shared__ unsigned pick_red[8];
    // The step's input row (into xb).  Called in layer 0's stage A when its weight rows have been requested: the pick is a chain of two
    // dependent round trips (the logits, then the picked id's table row) and the weights travel under it instead of behind it.
    __shared__ int pk_wsum[2][8];
    __shared__ int pk_isum[8], pk_pick;
    auto load_step_input = [&]() {
    if (a.pick_logits && a.pick_top_k > 0) {
        // The previous step's codebook SAMPLED here (the reference's default for Qwen3-TTS is top-k 50 at temperature 0.9, qwen3_tts.py:1088:
        // through sampler launches between the steps a sampled one-request frame cost 17 % more than a greedy one).  k_sample_topk's contract
        // and mechanics on 512 threads x 4 entries of the 2048-entry row: keys in registers, bisection on the key value with a block-wide
        // count per probe, ties in index order through one packed prefix scan,
        // rank sort of the k candidates, exp2 / sums in order, one Philox draw — every block for itself (same bits everywhere).
        // LDS: the candidate arrays alias hb (not in use before stage C of layer 0).
        u32* cand_key = reinterpret_cast<u32*>(hb);                 // [256] each: 5 KiB of hb's 6
        int* cand_idx = reinterpret_cast<int*>(cand_key + 256);
        u32* srt_key = reinterpret_cast<u32*>(cand_idx + 256);
        int* srt_idx = reinterpret_cast<int*>(srt_key + 256);
        float* pe = reinterpret_cast<float*>(srt_idx + 256);
        const int k = a.pick_top_k < 2048 ? a.pick_top_k : 2048;
        u32 key[4];
        {
            const uint2 w = reinterpret_cast<const uint2*>(a.pick_logits)[tid];
            const bf16_t raw[4] = {(bf16_t)(w.x & 0xffff), (bf16_t)(w.x >> 16), (bf16_t)(w.y & 0xffff), (bf16_t)(w.y >> 16)};
#pragma unroll
            for (int i = 0; i < 4; ++i) key[i] = key_of(f2bf(bf2f(raw[i]) / a.pick_temperature));
        }
        int par = 0;
        auto count_ge = [&](u32 t) {      // (a ballot + popcount per register: scalar arithmetic, no cross-lane moves)
            int c = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) c += __popcll(__ballot(key[i] >= t));
            if (lane == 0) pk_wsum[par][wave] = c;
            __syncthreads();
            int tot = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += pk_wsum[par][w];
            par ^= 1;
            return tot;
        };
        u32 lo = 0, hi = 65536;
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (count_ge(mid) >= k) lo = mid; else hi = mid;
        }
        const u32 T = lo;
        int ties = 0, aboves = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ties += key[i] == T ? 1 : 0; aboves += key[i] > T ? 1 : 0; }
        const int mine = ties | (aboves << 16);
        int inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, VOX_WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == 63) pk_isum[wave] = inc;
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { base += w < wave ? pk_isum[w] : 0; total += pk_isum[w]; }
        const int excl = base + inc - mine;
        const int tie_before = excl & 0xffff, above_before = excl >> 16, r = k - (total >> 16);
        int slot = above_before + (tie_before < r ? tie_before : r), tb = tie_before;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool isT = key[i] == T;
            if (key[i] > T || (isT && tb < r)) { cand_key[slot] = (key[i] << 16) | (0xFFFFu - (u32)(4 * tid + i)); ++slot; }      // (one word in candidate order: k_sample_topk)
            tb += isT ? 1 : 0;
        }
        __syncthreads();
        for (int j = tid; j < k; j += 512) {
            const u32 cj = cand_key[j];
            int rank = 0;
            for (int i = 0; i < k; ++i) rank += cand_key[i] > cj ? 1 : 0;
            srt_key[rank] = cj >> 16;
            srt_idx[rank] = (int)(0xFFFFu - (cj & 0xFFFFu));
        }
        __syncthreads();
        const float mval = bf2f(bits_of(srt_key[0]));
        for (int j = tid; j < k; j += 512) pe[j] = exp2_c((bf2f(bits_of(srt_key[j])) - mval) * VOX_LOG2E);
        __syncthreads();
        if (wave == 0) {
            const uint64_t off = a.pick_offset + (a.pick_offset_dev ? (*a.pick_offset_dev) * a.pick_offset_mul : 0ull);
            const float u = (float)(philox_u32(a.pick_seed, off, 0u) >> 8) * (1.0f / 16777216.0f);
            const int pick = sample_tail_wave(pe, k, a.pick_min_p, a.pick_top_p, u, lane);
            if (lane == 0) pk_pick = srt_idx[pick];
        }
        __syncthreads();
        const int picked = pk_pick;
        reinterpret_cast<unsigned*>(xb)[tid] = reinterpret_cast<const unsigned*>(a.pick_tab + (size_t)picked * H)[tid];
        if (blk == 0) {
            if (tid == 0) *a.pick_out = picked;
            const uint4* src = reinterpret_cast<const uint4*>(a.pick_emb + (size_t)picked * a.pick_H);
            uint4* fa = reinterpret_cast<uint4*>(a.pick_feat);
            for (int i = tid; i < (a.pick_H >> 3); i += 512) {
                const uint4 e = src[i];
                const uint4 f = a.pick_init ? make_uint4(0, 0, 0, 0) : fa[i];
                uint4 o;
                o.x = (u32)f2bf(bflo(f.x) + bflo(e.x)) | ((u32)f2bf(bfhi(f.x) + bfhi(e.x)) << 16);
                o.y = (u32)f2bf(bflo(f.y) + bflo(e.y)) | ((u32)f2bf(bfhi(f.y) + bfhi(e.y)) << 16);
                o.z = (u32)f2bf(bflo(f.z) + bflo(e.z)) | ((u32)f2bf(bfhi(f.z) + bfhi(e.z)) << 16);
                o.w = (u32)f2bf(bflo(f.w) + bflo(e.w)) | ((u32)f2bf(bfhi(f.w) + bfhi(e.w)) << 16);
                fa[i] = o;
            }
        }
    } else if (a.pick_logits) {
        // The previous step's codebook, picked here instead of by a sampler launch in between (greedy frames): every block takes the
        // first maximum of the 2048 logits — order of (value, lowest index), the sampler's — and reads the step's input row of that id
        // from the tabulated projection; block 0 records the id and adds the id's embedding to the next frame's feature row.
        unsigned best = 0;
        auto take = [&](unsigned h16, int v) {
            bf16_t b = (bf16_t)h16;
            if (b == 0x8000) b = 0;                                   // -0 == +0
            const unsigned key = (b & 0x8000) ? (unsigned)(~b & 0xffff) : (unsigned)(b | 0x8000);
            const unsigned c = (key << 16) | (0xFFFFu - (unsigned)v);
            best = c > best ? c : best;
        };
        for (int c = tid; c < (a.pick_vocab >> 2); c += 512) {
            const uint2 q = reinterpret_cast<const uint2*>(a.pick_logits)[c];
            take(q.x & 0xffff, 4 * c); take(q.x >> 16, 4 * c + 1); take(q.y & 0xffff, 4 * c + 2); take(q.y >> 16, 4 * c + 3);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const unsigned o = __shfl_xor(best, off, VOX_WAVE); best = o > best ? o : best; }
        if (lane == 0) pick_red[wave] = best;
        __syncthreads();
        unsigned m = pick_red[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = pick_red[w] > m ? pick_red[w] : m;
        const int picked = (int)(0xFFFFu - (m & 0xFFFFu));
        reinterpret_cast<unsigned*>(xb)[tid] = reinterpret_cast<const unsigned*>(a.pick_tab + (size_t)picked * H)[tid];
        if (blk == 0) {
            if (tid == 0) *a.pick_out = picked;
            const uint4* src = reinterpret_cast<const uint4*>(a.pick_emb + (size_t)picked * a.pick_H);
            uint4* fa = reinterpret_cast<uint4*>(a.pick_feat);
            for (int i = tid; i < (a.pick_H >> 3); i += 512) {
                const uint4 e = src[i];
                const uint4 f = a.pick_init ? make_uint4(0, 0, 0, 0) : fa[i];
                uint4 o;
                o.x = (u32)f2bf(bflo(f.x) + bflo(e.x)) | ((u32)f2bf(bfhi(f.x) + bfhi(e.x)) << 16);
                o.y = (u32)f2bf(bflo(f.y) + bflo(e.y)) | ((u32)f2bf(bfhi(f.y) + bfhi(e.y)) << 16);
                o.z = (u32)f2bf(bflo(f.z) + bflo(e.z)) | ((u32)f2bf(bfhi(f.z) + bfhi(e.z)) << 16);
                o.w = (u32)f2bf(bflo(f.w) + bflo(e.w)) | ((u32)f2bf(bfhi(f.w) + bfhi(e.w)) << 16);
                fa[i] = o;
            }
        }
    } else {
        // x of the step: plain row written by the previous kernel
        reinterpret_cast<unsigned*>(xb)[tid] = reinterpret_cast<const unsigned*>(a.x_in)[tid];
    }
    };
#if !VOX_DS_PICK_LATE
    load_step_input();
#endif
    for (int l = 0; l < a.n_layers; ++l) {
        // (constant address space: scalar loads straight into scalar registers — as plain loads the table row came through vector registers, a
        // round trip and nine readfirstlanes in front of the layer's first weight request)
        DepthLayerW w;
        {
            static_assert(sizeof(DepthLayerW) == 9 * sizeof(void*), "DepthLayerW: nine pointers");
            const VOX_CONST_AS unsigned long long* lp = (const VOX_CONST_AS unsigned long long*)(a.layers + l);
            const bf16_t** wp = reinterpret_cast<const bf16_t**>(&w);
#pragma unroll
            for (int i = 0; i < 9; ++i) wp[i] = assume_global(reinterpret_cast<const bf16_t*>(lp[i]));
        }
// Where the depth step asks for the layer's cached K / V rows (all bit-identical; round-6 A/B, one-request frame):  0 the general path in front
// of the qkv gather (round 5) | 1 the compile-time path there, nothing else: 2.41 ms — the gather's polls start ~1 us sooner and polling early
// is traffic | 2 ... + a wait for the rows before the first poll: 2.34 | 5 with stage A's weight rows: 66 spilled registers, 2.66 | 6 K in
// front of the gather, V behind it, no wait: 2.27 | 7 (default) K + wait in front, V behind the gather (needed a microsecond later than K;
// in front, their 15 requests per wave stood between the block and its polls): -0.6 % against 2 with the qkv delay retuned (4 -> 6).
#ifndef VOX_DS_KV_MODE
#define VOX_DS_KV_MODE 7
#endif
        // the layer's attention operands that do not depend on this launch (cached K / V of the earlier tokens, norm weights, RoPE entries)
        AttnArgs at = a.at;
        at.kv = a.at.kv + (size_t)l * a.kv_layer_stride;
        at.kv_w = a.at.kv_w + (size_t)l * a.kv_layer_stride;
        at.qn = w.qn; at.kn = w.kn;
        AttnShortPre<NT> pf;
        // ---------------- stage A: qkv = Wqkv . rmsnorm(x, ln1)  (2048 column pairs: one per wave of every block) ----------------
        {
            const int pr = blk * 8 + wave, n0 = 2 * pr;
            uint4 wq[2][2], nwv[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint4* wr = reinterpret_cast<const uint4*>(w.wqkv + (size_t)(n0 + r) * H);
#pragma unroll
                for (int j = 0; j < 2; ++j) wq[r][j] = ldg(wr + lane + 64 * j);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) nwv[j] = ldg(reinterpret_cast<const uint4*>(w.ln1) + lane + 64 * j);
            // (every wave requests — waves 2..7 the rows of wave 0 / 1 again, cache hits — and the chunks are named values, not an array: as
            // an array filled under `wave < 2` they were kept in scratch memory, stored behind a vmcnt(0) each)
            const uint4* wor0 = reinterpret_cast<const uint4*>(w.wo + (size_t)(2 * (blk * 2 + (wave & 1))) * NQ) + lane;
            const uint4* wor1 = wor0 + NQ / 8;
            const uint4 wo00 = ldg(wor0), wo01 = ldg(wor0 + 64), wo02 = ldg(wor0 + 128), wo03 = ldg(wor0 + 192);
            const uint4 wo10 = ldg(wor1), wo11 = ldg(wor1 + 64), wo12 = ldg(wor1 + 128), wo13 = ldg(wor1 + 192);
#if VOX_DS_KV_MODE == 5
            // requested HERE, behind the stage's weight rows and in front of the x gather: between stage A's publish and the qkv gather (where
            // they used to go out) their return stood in front of the gather's polls — loads return in order — for 0.1 us per visible token
            attn_short_prefetch<NT, true>(at, 0, wave, lane, pf);
#endif
            if (l > 0) gran_gather_lds<512>(a.gx, tagof(l - 1, 3), reinterpret_cast<unsigned*>(xb), tid, a.err, 0x100u + l, max_spins, a.poll_delay & 255u);
            if (l == 0) {
#if VOX_DS_PICK_LATE
                load_step_input();
#endif
                // the launch's state words: plain loads of a uniform address (scalar loads: the words were written before this launch started —
                // the previous launch's block 0, or the host — and a kernel boundary stands in between), BEHIND the step's first weight
                // requests: as L1-bypassing vector loads at the top of the kernel they were three dependent round trips in front of them
                asm volatile("" ::: "memory");
                const VOX_CONST_AS unsigned* wp = (const VOX_CONST_AS unsigned*)a.epoch;      // (constant address space: scalar loads)
                ep = wp[0]; max_spins = wp[2]; drop_first = blk == 1 && wp[3] != 0u; tag0 = ep * 64u;
            }
            __syncthreads();                               // x of this layer is in xb
            uint4 xv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) xv[j] = xb[lane + 64 * j];
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) s = sq8(xv[j], s);
            s = butterfly<64>(s);
            const float rinv = 1.0f / sqrtf(s / (float)H + a.eps);
#pragma unroll
            for (int j = 0; j < 2; ++j) xv[j] = norm_chunk(xv[j], nwv[j], rinv);
            float acc[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float d = 0.0f;
#pragma unroll
                for (int j = 0; j < 2; ++j) d = dot8(wq[r][j], xv[j], d);
                acc[r] = butterfly<64>(d);
            }
            if (lane == 0 && !(drop_first && l == 0)) gran_write(a.gqkv + pr, tagof(l, 0), f2bf(acc[0]), f2bf(acc[1]));
            __builtin_amdgcn_sched_barrier(0);      // (stage B's requests stay below: hoisted above the parking they made the allocator spill these rows)
            if (wave < 2) {
                wos[wave][0][0][lane] = wo00; wos[wave][0][1][lane] = wo01; wos[wave][0][2][lane] = wo02; wos[wave][0][3][lane] = wo03;
                wos[wave][1][0][lane] = wo10; wos[wave][1][1][lane] = wo11; wos[wave][1][2][lane] = wo12; wos[wave][1][3][lane] = wo13;
            }
            __builtin_amdgcn_sched_barrier(0);
            VOX_STAMP2(1 + 6 * l)
        }
        // ---------------- stage B: x += Wo . attention  (512 pairs: waves 0, 1; the attention by all 8 waves = 8 kv heads) ----------------
        {
            const int pr = blk * 2 + wave;
            const int hk = wave;
#if VOX_DS_KV_MODE == 6 || VOX_DS_KV_MODE == 7
            attn_short_prefetch<NT, true, 1>(at, 0, hk, lane, pf);
#elif VOX_DS_KV_MODE != 5
            attn_short_prefetch<NT, VOX_DS_KV_MODE != 0>(at, 0, hk, lane, pf);
#endif
#if VOX_DS_KV_MODE == 2 || VOX_DS_KV_MODE == 7
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif VOX_DS_KV_MODE == 3
            __builtin_amdgcn_s_sleep(32);
#elif VOX_DS_KV_MODE == 4
            __builtin_amdgcn_s_sleep(64);
#endif
            gran_gather_lds<2048>(a.gqkv, tagof(l, 0), reinterpret_cast<unsigned*>(qb), tid, a.err, 0x200u + l, max_spins, (a.poll_delay >> 8) & 255u);
#if VOX_DS_KV_MODE == 6 || VOX_DS_KV_MODE == 7
            attn_short_prefetch<NT, true, 2>(at, 0, hk, lane, pf);
#endif
            __syncthreads();                               // q | k | v in qb (and: every wave is done with xb's stage-A reads)
            VOX_STAMP2(2 + 6 * l)
            {
                // this lane's chunk of its group's head row (q head 2 hk / 2 hk + 1, new k, new v) and its 4 dims of the new v
                const int grp = lane >> 4, j16 = lane & 15, dq = lane & 31;
                const int col = (grp == 0 ? (hk * 2) * 128 : grp == 1 ? (hk * 2 + 1) * 128 : grp == 2 ? NQ + hk * 128 : NQ + 1024 + hk * 128) + 8 * j16;
                const uint4 v = qb[col >> 3];
                const uint2 vnew = reinterpret_cast<const uint2*>(qb)[(NQ + 1024 + hk * 128 + 4 * dq) >> 2];
                attn_short_compute<NT>(at, 0, hk, lane, v, vnew, pf, reinterpret_cast<bf16_t*>(xs), blk == 0);
            }
            __syncthreads();                               // the attention row is in xs
            VOX_STAMP2(3 + 6 * l)
            if (wave < 2) {
                const unsigned resw = reinterpret_cast<const unsigned*>(xb)[pr];     // this pair's x (xb still holds the layer's input)
                float acc[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    float d = 0.0f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) d = dot8(wos[wave][r][j][lane], xs[lane + 64 * j], d);
                    acc[r] = butterfly<64>(d);
                }
                if (lane == 0) {
                    const bf16_t r0 = f2bf(bflo(resw) + bf2f(f2bf(acc[0]))), r1 = f2bf(bfhi(resw) + bf2f(f2bf(acc[1])));
                    gran_write(a.gx + pr, tagof(l, 1), r0, r1);
                }
            }
            VOX_STAMP2(4 + 6 * l)
        }
        // ---------------- stage C: h = silu(Wg . n) * (Wu . n), n = rmsnorm(x, ln2)  (1536 pairs: waves 0..5) ----------------
        {
            const int pr = blk * 6 + wave, f0 = 2 * pr;
            uint4 wg[2][2], wu[2][2], nwv[2];
            if (wave < 6) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint4* gr = reinterpret_cast<const uint4*>(w.wgate + (size_t)(f0 + r) * H);
                    const uint4* ur = reinterpret_cast<const uint4*>(w.wup + (size_t)(f0 + r) * H);
#pragma unroll
                    for (int j = 0; j < 2; ++j) { wg[r][j] = ldg(gr + lane + 64 * j); wu[r][j] = ldg(ur + lane + 64 * j); }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) nwv[j] = ldg(reinterpret_cast<const uint4*>(w.ln2) + lane + 64 * j);
            }
            // (xc's last readers, the previous layer's stage-D residual words, were done before this layer's stage-A barrier: the barrier
            // below is not needed for correctness — it parks the waves that have nothing to do in stage B's o_proj away from the poll loop)
            if (VOX_DS_EXTRA_BARRIERS) __syncthreads();
            gran_gather_lds<512>(a.gx, tagof(l, 1), reinterpret_cast<unsigned*>(xc), tid, a.err, 0x400u + l, max_spins, (a.poll_delay >> 16) & 255u);
            __syncthreads();
            if (wave < 6) {
                uint4 xv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) xv[j] = xc[lane + 64 * j];
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < 2; ++j) s = sq8(xv[j], s);
                s = butterfly<64>(s);
                const float rinv = 1.0f / sqrtf(s / (float)H + a.eps);
#pragma unroll
                for (int j = 0; j < 2; ++j) xv[j] = norm_chunk(xv[j], nwv[j], rinv);
                bf16_t hv[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    float dg = 0.0f, du = 0.0f;
#pragma unroll
                    for (int j = 0; j < 2; ++j) dg = dot8(wg[r][j], xv[j], dg);
#pragma unroll
                    for (int j = 0; j < 2; ++j) du = dot8(wu[r][j], xv[j], du);
                    const float gg = bfround(butterfly<64>(dg)), uu = bfround(butterfly<64>(du));
                    hv[r] = f2bf(bfround(silu_c(gg)) * uu);
                }
                if (lane == 0) gran_write(a.gh + pr, tagof(l, 2), hv[0], hv[1]);
            }
            VOX_STAMP2(5 + 6 * l)
        }
        // ---------------- stage D: x += Wd . h  (512 pairs: waves 0, 1) ----------------
        {
            const int pr = blk * 2 + wave, n0 = 2 * pr;
            uint4 wd[2][6];
            if (wave < 2) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint4* wr = reinterpret_cast<const uint4*>(w.wdown + (size_t)(n0 + r) * F);
#pragma unroll
                    for (int j = 0; j < 6; ++j) wd[r][j] = ldg(wr + lane + 64 * j);
                }
            }
            gran_gather_lds<1536>(a.gh, tagof(l, 2), reinterpret_cast<unsigned*>(hb), tid, a.err, 0x600u + l, max_spins, a.poll_delay >> 24);
            __syncthreads();                               // h in hb (xc = x after attention: the residual of this stage)
            if (wave < 2) {
                const unsigned resw = reinterpret_cast<const unsigned*>(xc)[pr];
                float acc[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    float d = 0.0f;
#pragma unroll
                    for (int j = 0; j < 6; ++j) d = dot8(wd[r][j], hb[lane + 64 * j], d);
                    acc[r] = butterfly<64>(d);
                }
                if (lane == 0) {
                    const bf16_t r0 = f2bf(bflo(resw) + bf2f(f2bf(acc[0]))), r1 = f2bf(bfhi(resw) + bf2f(f2bf(acc[1])));
                    gran_write(a.gx + pr, tagof(l, 3), r0, r1);
                }
            }
            // (the next layer's stage A gathers into xb, whose last readers — stage A's rows and stage B's residual words of THIS layer —
            // were done before the stage-C barrier, and hb is rewritten only after three more barriers: this barrier, too, only keeps
            // the six idle waves out of the next poll loop while waves 0, 1 finish the down projection)
            if (VOX_DS_EXTRA_BARRIERS) __syncthreads();
            VOX_STAMP2(6 + 6 * l)
        }
    }
    // ---------------- head: logits = Whead . rmsnorm(x, final_norm)  (1024 pairs: waves 0..3; plain stores, a kernel boundary follows) ----------------
    {
        const int pr = blk * 4 + wave, n0 = 2 * pr;
        uint4 wh[2][2], nwv[2];
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint4* wr = reinterpret_cast<const uint4*>(a.head_w + (size_t)(n0 + r) * H);
#pragma unroll
                for (int j = 0; j < 2; ++j) wh[r][j] = wr[lane + 64 * j];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) nwv[j] = reinterpret_cast<const uint4*>(a.final_norm)[lane + 64 * j];
        }
        gran_gather_lds<512>(a.gx, tagof(a.n_layers - 1, 3), reinterpret_cast<unsigned*>(xb), tid, a.err, 0x700u, max_spins, a.poll_delay & 255u);
        __syncthreads();
        if (wave < 4) {
            uint4 xv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) xv[j] = xb[lane + 64 * j];
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) s = sq8(xv[j], s);
            s = butterfly<64>(s);
            const float rinv = 1.0f / sqrtf(s / (float)H + a.eps);
#pragma unroll
            for (int j = 0; j < 2; ++j) xv[j] = norm_chunk(xv[j], nwv[j], rinv);
            float acc[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float d = 0.0f;
#pragma unroll
                for (int j = 0; j < 2; ++j) d = dot8(wh[r][j], xv[j], d);
                acc[r] = butterfly<64>(d);
            }
            if (lane == 0) reinterpret_cast<unsigned*>(a.logits)[pr] = (unsigned)f2bf(acc[0]) | ((unsigned)f2bf(acc[1]) << 16);
        }
    }
    VOX_STAMP2(31)
    if (blk == 0 && tid == 0) __hip_atomic_store(a.epoch, ep + 1u, VOX_RLX_AGENT);
    if (drop_first && tid == 0) atomicSub(a.epoch + 3, 1u);
}

// ================================================================================================
// Persistent MLP half of a talker layer (one request): o_proj + residual, gate/up + SiLU*up, down + residual — three dependent
// weight-streaming stages (8.4 + 50.3 + 25.2 MB at Qwen3-TTS-1.7B) — in ONE launch of 256 resident blocks with the same granule
// hand-offs as k_depth_step.  What it buys over three launches: the next stage's weights are requested BEFORE the hand-off is
// polled (waves with no work in the first stage request theirs when the kernel starts), so HBM keeps streaming across the two
// seams instead of idling for a kernel boundary plus a first-load latency each.  Arithmetic: k_gemv<1, 4 / 12, ...> stage for
// stage — same lane -> chunk assignment, dot8 / butterfly<64> order and rounding points: bit-identical to the launch chain.
//   stage O  x'  = x + Wo . attn                                  1024 column pairs: waves 0..3
//   stage C  h   = silu(Wg . n) * (Wu . n),  n = rmsnorm(x', ln2)  3072 pairs: every wave one, waves 0..3 a second one
//   stage D  x'' = x' + Wd . h                                    1024 pairs: waves 4..7 (round 6: rows requested when x' is gathered)
//   stage A  qkv = Wqkv' . rmsnorm(x'', ln1')                      the NEXT layer's projection, 2048 pairs: every wave one
// Round 6: the layer's decode attention in blocks 0..15 of the launch (ATTN, <= 256 visible tokens), and then every layer of the stack
// in the launch (MULTI: a do-while over the layer table; q | k | v reach the next layer's attention blocks as granules).
// ================================================================================================
// The one-launch decode attention (k_attn_decode8's arithmetic, chunk by chunk and merge, bit for bit) as a device function for a
// block of NT threads inside a persistent kernel: NG groups of NT / NG threads (512 threads, 8 chunks: one wave per chunk), LDS in
// the caller's AttnDecodeSmem, the result published as hand-off granules instead of stored.  Per (row, q head) the order of every
// sum is the launch's own: which threads carry a chunk does not enter the arithmetic.
// ================================================================================================
template <int D, int GMAX, int NG, int NCH, int NT>
struct AttnDecodeSmem {
    static constexpr int LPT = D / 8;
    uint4 Ks[NG][VOX_TC * LPT];
    uint4 Vs[NG][VOX_TC * LPT];
    uint4 Qs[GMAX * LPT];
    float S[NG][GMAX][VOX_TC];
    float Ms[NG][GMAX];
    float Sh[(NT / 64) * D];
    bf16_t Knew[D];
    float Po[NCH][GMAX][D];
    float2 Pml[NCH][GMAX];
    float Wm[NCH][GMAX];
    bf16_t Raw[3 * D];          // RAWLDS: q head | k head | v head of the new token, staged by the caller's `stage_raw`
};
// RAWLDS (GMAX = 1): the new token's q / k / v rows are not read from a.qkv but from sm.Raw, which `stage_raw()` fills (it ends with a
// block barrier) — called BEHIND the first tile's K / V requests, so that the cached rows travel while the caller still waits for q | k | v
struct NoStage { __device__ void operator()() const {} };
template <int D, int GMAX, int NG, int NCH, int NT, bool RAWLDS = false, typename AfterPark, typename StageRaw = NoStage>
__device__ __forceinline__ void attn_decode8_block(const AttnArgs& a, AttnDecodeSmem<D, GMAX, NG, NCH, NT>& sm, int HS, int hk, int hs, int row,
                                                    unsigned long long* gout, const unsigned* gtag_lds, bool skip_publish, AfterPark after_park,
                                                    const unsigned* dbg_words = nullptr, StageRaw stage_raw = StageRaw()) {
    constexpr bool kRawLds = RAWLDS;
    static_assert(!RAWLDS || GMAX == 1, "RAWLDS: one q head per block");
    constexpr int LPT = D / 8, TPW = 64 / LPT, GT = NT / NG, GW = GT / 64, CPG = NCH / NG;
    constexpr int KVL = (VOX_TC * LPT + GT - 1) / GT;
    constexpr bool QREG = GMAX <= 2;
    static_assert(GT % 64 == 0 && GW >= 1, "a group is whole waves");
    VOX_STAMP2_DECL
    VOX_STAMP2(0)
#ifdef VOX_DEV_KNOBS
    if (stamp2_base) {      // latency probes at launch start (stamped wave only): a plain load, then an L1-bypassing load of the epoch word
        const int p0 = a.q_kvlen[row];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp2_base[11] = wall_clock64() + (p0 < -5);
        const unsigned w0 = __hip_atomic_load(gtag_lds == nullptr ? (const unsigned*)a.pos : dbg_words, VOX_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp2_base[12] = wall_clock64() + (w0 == 0x7fffffffu);
        const int p1 = a.ptab ? a.ptab[0] : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp2_base[13] = wall_clock64() + (p1 < -5);
    }
#endif
    int tid_ = threadIdx.x;
    if constexpr (RAWLDS) asm volatile("" : "+v"(tid_));      // (called once per layer of a loop: nothing per-thread hoisted out of it)
    const int tid = tid_, lane = tid & 63, wave16 = tid >> 6;
    const int grp = tid / GT, gt = tid % GT, gw = wave16 % GW;
    const int* pages = a.identity_pages ? nullptr
                       : (a.ptab ? a.ptab + (size_t)row * a.pt_stride : a.indices + a.indptr[a.q_req[row]]);
    // per-row page table: the page ids of this thread's tokens do not depend on the row's length, so they are requested
    // together with it (one exposed round trip in front of the K/V loads instead of two); entries past the row's last page are
    // read (clamped to the table row) and never used
    int pgi_pre[CPG][KVL];
    const bool hoist = a.ptab && a.hoist;
    const bool one_page = a.page_shift >= 5 && (GT % LPT) == 0;      // (VOX_TC = 32 tokens: a chunk never straddles a page; VOX_KV_ASM_FETCH)
    if (hoist) {
#pragma unroll
        for (int ci = 0; ci < CPG; ++ci)
#pragma unroll
            for (int u = 0; u < KVL; ++u) {
                if (one_page && u > 0) { pgi_pre[ci][u] = 0; continue; }      // one page id per chunk
                const int tok = (grp + NG * ci) * VOX_TC + (gt + GT * u) / LPT;
                const int pi = page_of(a, tok);
                pgi_pre[ci][u] = pages[pi < a.pt_stride ? pi : a.pt_stride - 1];
            }
    }
    const int L = a.fixed_kvlen > 0 ? a.fixed_kvlen : a.q_kvlen[row];
    // where the new token goes in the paged cache: requested HERE, with the row's length.  Read where it is used (behind the caller's
    // after_park) the two words were vector loads in the own-last-chunk branch, waited for with vmcnt(0) — i.e. behind the o_proj weight
    // rows after_park had just requested (loads return in order): the score pass of the in-launch attention took 3.3 us instead of 1.1
    const int pg_new = a.identity_pages ? row : a.page[row];
    const int sl_new = a.identity_pages ? 0 : a.slot[row];
#ifdef VOX_DEV_KNOBS
    if (stamp2_base) { stamp2_base[14] = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp2_base[15] = wall_clock64() + (L < -5); }
#endif
    const int nc = (L + VOX_TC - 1) / VOX_TC;                 // 1..NCH
    const int Gf = a.Hq / a.Hkv, G = Gf / HS, g0 = hs * G;
    const size_t ps = (size_t)2 * a.page_size * a.Hkv * D;
    const int nqkv = (a.Hq + 2 * a.Hkv) * D;
    const bf16_t* raw = a.qkv + (size_t)row * nqkv;

    // K/V tiles of this group's chunks.  The row's newest token (index L - 1) comes from the projection output: its V row is
    // loaded into the tile here, its K row (per-head norm + RoPE below) is read from Knew by the score pass.
#if VOX_KV_ASM
    VOX_KV_ASM_FETCH
#else
    uint4 kreg[KVL], vreg[KVL];
    auto fetch_tile = [&](int ci) {
        const int t0 = (grp + NG * ci) * VOX_TC;
#pragma unroll
        for (int u = 0; u < KVL; ++u) {
            const int i = gt + GT * u, t = i / LPT, j = i % LPT;
            kreg[u] = make_uint4(0, 0, 0, 0);
            vreg[u] = kreg[u];
            const int tok = t0 + t;
            if (i < VOX_TC * LPT && tok < L - 1) {
                const int pgi = hoist ? pgi_pre[ci][u] : (pages ? pages[page_of(a, tok)] : row);
                const bf16_t* base = a.kv + (size_t)pgi * ps + ((size_t)slot_of(a, tok) * a.Hkv + hk) * D;
                kreg[u] = reinterpret_cast<const uint4*>(base)[j];
                vreg[u] = reinterpret_cast<const uint4*>(base + (size_t)a.page_size * a.Hkv * D)[j];
            } else if (i < VOX_TC * LPT && tok == L - 1) {
                vreg[u] = reinterpret_cast<const uint4*>(raw + (size_t)(a.Hq + a.Hkv) * D + (size_t)hk * D)[j];
            }
        }
    };
    auto kv_wait = [&]() {};
    auto pages_arrived = [&]() {};
#endif
    // q heads of this kv head (per-head norm + RoPE) and the new k: one head per wave; the first head's operands are requested in
    // front of the tile
    int p = a.fixed_pos >= 0 ? a.fixed_pos : a.pos[row];
    p = p < 0 ? 0 : (p >= a.table_max_pos ? a.table_max_pos - 1 : p);
    const float* cs_row = a.cs ? a.cs + (size_t)p * (a.rot >> 1) * 2 : nullptr;
    auto head_src = [&](int h) { return h == G ? raw + (size_t)a.Hq * D + (size_t)hk * D : raw + (size_t)(hk * Gf + g0 + h) * D; };
    PrepOps po0;
    pages_arrived();
    if constexpr (RAWLDS) {
        // (the tile is waited for right here: its requests are asm statements — the compiler does not know their destinations are in
        // flight, and anything it might do with those registers in between, a copy at a loop edge or a spill, would read them too early)
        fetch_tile(0);
        kv_wait();
        VOX_STAMP2(1)
        if constexpr (CPG == 1) {
            // one chunk per group: the cached tile is parked HERE, while q | k | v are still on their way (0.64 us of the chain behind
            // their arrival before); the new token's V row is patched in below, when sm.Raw holds it
#pragma unroll
            for (int u = 0; u < KVL; ++u) {
                const int i = gt + GT * u;
                if (i < VOX_TC * LPT) { sm.Ks[grp][i] = as_uint4(kreg[u]); sm.Vs[grp][i] = as_uint4(vreg[u]); }
            }
        }
        stage_raw();                               // q | k | v of the new token -> sm.Raw (ends with a barrier)
        VOX_STAMP2(10)
        if (wave16 < 2) po0 = prep_head_load_lds<D>(sm.Raw + (wave16 == 1 ? D : 0), wave16 == 1 ? a.kn : a.qn, cs_row, a.rot, lane);
        if (wave16 < 2) {
            prep_head_arrived(po0);
            prep_head_apply<D>(po0, wave16 == 1 ? a.kn : a.qn, a.eps, cs_row, a.rot, a.interleave, sm.Sh + wave16 * D,
                               wave16 == 1 ? sm.Knew : reinterpret_cast<bf16_t*>(sm.Qs), lane);
        }
    } else {
    if (wave16 < G + 1) po0 = prep_head_load<D>(head_src(wave16), wave16 == G ? a.kn : a.qn, cs_row, a.rot, lane);
    fetch_tile(0);
    kv_wait();
    if (wave16 < G + 1) prep_head_arrived(po0);
    VOX_STAMP2(1)
    for (int h = wave16; h < G + 1; h += NT / 64) {
        const bool isk = h == G;
        bf16_t* dst = isk ? sm.Knew : reinterpret_cast<bf16_t*>(sm.Qs) + (size_t)h * D;
        if (h == wave16) prep_head_apply<D>(po0, isk ? a.kn : a.qn, a.eps, cs_row, a.rot, a.interleave, sm.Sh + wave16 * D, dst, lane);
        else prep_head<D>(head_src(h), isk ? a.kn : a.qn, a.eps, cs_row, a.rot, a.interleave, sm.Sh + wave16 * D, dst, lane);
    }
    }
    VOX_STAMP2(2)
#pragma unroll
    for (int ci = 0; ci < CPG; ++ci) {
        const int c = grp + NG * ci, t0 = c * VOX_TC;
        const bool live = t0 < L;
        const int nt = live ? ((L - t0) < VOX_TC ? (L - t0) : VOX_TC) : 0;
        const bool own_last = live && (t0 + nt == L);
        if (ci > 0) __syncthreads();           // the previous chunk's tile is dead
        if constexpr (RAWLDS && CPG == 1) {
            if (own_last && gt < LPT) sm.Vs[grp][(nt - 1) * LPT + gt] = reinterpret_cast<const uint4*>(sm.Raw + 2 * D)[gt];      // (the tile itself was parked above)
        } else {
#pragma unroll
        for (int u = 0; u < KVL; ++u) {
            const int i = gt + GT * u;
            if (i < VOX_TC * LPT) {
                sm.Ks[grp][i] = as_uint4(kreg[u]);
                uint4 vv = as_uint4(vreg[u]);
                if (RAWLDS && own_last && i / LPT == nt - 1) vv = reinterpret_cast<const uint4*>(sm.Raw + 2 * D)[i % LPT];
                sm.Vs[grp][i] = vv;
            }
        }
        }
        if (ci + 1 < CPG) fetch_tile(ci + 1);  // in flight during this chunk's arithmetic
        __syncthreads();                       // tiles parked; (ci = 0) Qs / Knew written
        if (ci == 0) after_park();             // the caller's own loads: nothing of the attention waits behind them any more
        VOX_STAMP2(3)
        if (own_last && hs == 0 && gt < LPT) {
            // append the new token to the paged cache (page < 0: graph padding row)
            const int pg = pg_new;
            const int sl = a.identity_pages ? (L - 1) : sl_new;
            if (pg >= 0) {
                bf16_t* base = a.kv_w + (size_t)pg * ps + ((size_t)sl * a.Hkv + hk) * D;
                reinterpret_cast<uint4*>(base)[gt] = reinterpret_cast<const uint4*>(sm.Knew)[gt];
                reinterpret_cast<uint4*>(base + (size_t)a.page_size * a.Hkv * D)[gt] = sm.Vs[grp][(nt - 1) * LPT + gt];
            }
        }
        if (live) {      // scores: LPT lanes per token, butterfly over LPT lanes; the token's K chunk is unpacked once for all q heads
            const int j = lane % LPT;
            float qf[QREG ? GMAX : 1][8];
            if constexpr (QREG) {
#pragma unroll
                for (int g = 0; g < GMAX; ++g) {
                    const uint4 qx = sm.Qs[(g < G ? g : 0) * LPT + j];
                    qf[g][0] = bflo(qx.x); qf[g][1] = bfhi(qx.x); qf[g][2] = bflo(qx.y); qf[g][3] = bfhi(qx.y);
                    qf[g][4] = bflo(qx.z); qf[g][5] = bfhi(qx.z); qf[g][6] = bflo(qx.w); qf[g][7] = bfhi(qx.w);
                }
            }
#pragma unroll
            for (int tb = gw * TPW; tb < VOX_TC; tb += GW * TPW) {
                const int tt = tb + lane / LPT;
                const uint4 kx = (own_last && tt == nt - 1) ? reinterpret_cast<const uint4*>(sm.Knew)[j] : sm.Ks[grp][tt * LPT + j];
                if constexpr (QREG) {
                    const float kf[8] = {bflo(kx.x), bfhi(kx.x), bflo(kx.y), bfhi(kx.y), bflo(kx.z), bfhi(kx.z), bflo(kx.w), bfhi(kx.w)};
#pragma unroll
                    for (int g = 0; g < GMAX; ++g) {
                        float d = 0.0f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) d = __fmaf_rn(qf[g][e], kf[e], d);
                        d = butterfly<LPT>(d);
                        if (j == 0 && g < G) sm.S[grp][g][tt] = d * a.scale;
                    }
                } else {
                    for (int g = 0; g < G; ++g) {
                        float d = dot8(sm.Qs[g * LPT + j], kx, 0.0f);
                        d = butterfly<LPT>(d);
                        if (j == 0) sm.S[grp][g][tt] = d * a.scale;
                    }
                }
            }
        }
        __syncthreads();
        VOX_STAMP2(4)
        if (live) {      // chunk max + p = exp2((s-m)*log2e): 32 lanes per q head
            for (int pr = gt; pr < G * VOX_TC; pr += GT) {
                const int g = pr / VOX_TC, t = pr % VOX_TC;
                const float s = t < nt ? sm.S[grp][g][t] : -INFINITY;
                float m = s;
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, VOX_WAVE));
                const float p = t < nt ? exp2_c((s - m) * VOX_LOG2E) : 0.0f;
                sm.S[grp][g][t] = p;
                if (t == 0) sm.Ms[grp][g] = m;
            }
        }
        __syncthreads();
        VOX_STAMP2(5)
        if (live) {      // PV: one thread per (q head, pair of dims); sequential over the tokens of the (zero-padded) tile
            const u32* Vw = reinterpret_cast<const u32*>(sm.Vs[grp]);
            for (int e = gt; e < G * (D / 2); e += GT) {
                const int g = e / (D / 2), dp = e % (D / 2);
                float o0 = 0.0f, o1 = 0.0f, l = 0.0f;
#pragma unroll(NCH == 8 ? VOX_TC : 8)                      // (beside an in-flight tile the full unroll's operand registers would spill)
                for (int t = 0; t < VOX_TC; ++t) {
                    const float p = sm.S[grp][g][t];          // 0 for t >= nt
                    const u32 vw = Vw[t * (D / 2) + dp];
                    l = l + p;
                    o0 = __fmaf_rn(p, bflo(vw), o0);
                    o1 = __fmaf_rn(p, bfhi(vw), o1);
                }
                *reinterpret_cast<float2*>(&sm.Po[c][g][2 * dp]) = make_float2(o0, o1);
                if (dp == 0) sm.Pml[c][g] = make_float2(sm.Ms[grp][g], l);
            }
        }
    }
    __syncthreads();
    VOX_STAMP2(6)
    // merge (k_attn_merge): global max, the chunk weights w_c = exp2((m_c - M) log2e) once per (chunk, head), then L and O over
    // the chunks in ascending order
    if (tid < NCH * GMAX) {
        const int c = tid / GMAX, g = tid % GMAX;
        if (g < G) {
            float M = -INFINITY;
            for (int cc = 0; cc < nc; ++cc) M = fmaxf(M, sm.Pml[cc][g].x);
            sm.Wm[c][g] = c < nc ? exp2_c((sm.Pml[c][g].x - M) * VOX_LOG2E) : 0.0f;
        }
    }
    __syncthreads();
    VOX_STAMP2(7)
    // output: q head (hk * Gf + g0 + g), two dims per thread -> one granule {bf16 d, bf16 d + 1, tag} (the MLP half of the
    // layer gathers the 2048-wide attention row from them: no store / re-load through a kernel boundary)
    const unsigned gtag = *gtag_lds;             // (written by the caller's after_park, barriers ago)
    for (int e = tid; e < G * (D / 2); e += NT) {
        const int g = e / (D / 2), dp = e % (D / 2);
        float Lsum = 0.0f, O0 = 0.0f, O1 = 0.0f;
        for (int c = 0; c < nc; ++c) {
            const float w = sm.Wm[c][g];
            Lsum = __fmaf_rn(sm.Pml[c][g].y, w, Lsum);
            O0 = __fmaf_rn(sm.Po[c][g][2 * dp], w, O0);
            O1 = __fmaf_rn(sm.Po[c][g][2 * dp + 1], w, O1);
        }
        const int h = hk * Gf + g0 + g;
        if (!skip_publish) gran_write(gout + ((size_t)row * a.Hq + h) * (D / 2) + dp, gtag, f2bf(O0 / Lsum), f2bf(O1 / Lsum));
    }
    VOX_STAMP2(8)
}

struct TalkerMlpArgs {
    const bf16_t *wo, *wgate, *wup, *wdown, *ln2;
    const bf16_t *wqkv_next, *ln1_next;      // the NEXT layer's input_layernorm + q/k/v projection as a fourth stage (NULL: last layer)
    bf16_t* qkv_out;                          // plain row [4096] for the attention launch that follows
    const bf16_t *attn, *x_in;       // plain rows [2048]
    bf16_t* x_out;                   // plain row [2048] (may alias x_in)
    unsigned long long *gx, *gh;     // granules: 1024, 3072
    unsigned *epoch, *err;
    float eps;
    // ATTN form (k_talker_mlp<true>): the layer's decode attention runs in blocks 0..15 of this launch (one q head each, <= 256 visible
    // tokens, k_attn_decode8's arithmetic) and reaches stage O as granules; `attn` is unused
    unsigned long long* gattn;       // granules: 1024 (the 2048-wide attention row)
    int burst_delay;                 // plain blocks: s_sleep(16) repeats in front of their first weight requests
    int poll_sleep;                  // s_sleep(8) repeats between two polls of the sentinel granule
    unsigned poll_delay;             // first-pass hold-back of the gathers, one byte each (x 128 clocks): [7:0] x' (O -> C), [15:8] h (C -> D), [23:16] x (D -> next qkv), [31:24] the attention row (ATTN form)
    AttnArgs at;
    // MULTI form (k_talker_mlp<ATTN, true>): ALL decoder layers in one launch.  Layer l takes its weights from tab[l] (and the next layer's
    // q/k/v projection from tab[l + 1]), its K / V cache at at.kv + l * kv_layer_stride; stage A hands q | k | v to the next layer's
    // attention blocks as granules (gq) instead of a plain row behind a kernel boundary.  Layer 0's q | k | v is the plain row at.qkv.
    const struct TalkerLayerW* tab;
    int n_layers;
    long kv_layer_stride;
    unsigned long long* gq;          // granules: 2048 (the 4096-wide q | k | v row)
};
struct TalkerLayerW { const bf16_t *wo, *wgate, *wup, *wdown, *ln2, *wqkv, *ln1, *qn, *kn; };
template <int TOTAL>
__device__ __forceinline__ void gran_gather_lds_all(const unsigned long long* g, unsigned tag, unsigned* dst, int tid, unsigned* err, unsigned code, unsigned max_spins, unsigned delay = 0) {
    constexpr int PER = (TOTAL + 511) / 512;
#if VOX_GRAN_ASM
    u32x2_t gv[PER];
    const unsigned long long* gp[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) gp[q] = g + (tid + 512 * q < TOTAL ? tid + 512 * q : tid);
    gran_poll_all<PER, TOTAL>(g, tid, gp, gv, tag, [&](int q) { return tid + 512 * q < TOTAL; }, err, code, max_spins, delay);
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (tid + 512 * q < TOTAL) dst[tid + 512 * q] = gv[q].x;
#else
    unsigned val[PER];
    for (unsigned spin = 0;; ++spin) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (tid + 512 * q < TOTAL) {
                const unsigned long long x = __hip_atomic_load(g + tid + 512 * q, VOX_RLX_AGENT);
                ok = ok && (unsigned)(x >> 32) == tag;
                val[q] = (unsigned)x;
            }
        }
        if (ok) break;
        if (spin > max_spins) { atomicCAS(err, 0u, code); break; }
        if ((spin & 63u) == 63u && __hip_atomic_load(err, VOX_RLX_AGENT) != 0u) break;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (tid + 512 * q < TOTAL) dst[tid + 512 * q] = val[q];
#endif
}

// ATTN: the whole decoder layer of a one-request frame in ONE launch — blocks 0..15 first run the decode attention of one q head each
// (their K/V tiles and page ids are requested when the launch starts: no kernel boundary and no second exposed round trip in front
// of them), while the other 240 blocks already hold their first o_proj / gate / up weight rows in registers: the weight stream of the
// MLP half runs under the attention's latency chain instead of behind it.
template <int NCH> using TalkerAttnSmemT = AttnDecodeSmem<128, 1, 8, (NCH ? NCH : 8), 512>;
#ifdef VOX_DEV_KNOBS
// development builds: phase stamps of the layer launch from an attention block (0) and a plain block (100): records {kind 4 | 5, ATTN, t0..t7, end}
#define MLP_TR_DECL unsigned long long mt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mslot = ~0ull;                                      \
    const bool mtr = g_vox_trace && tid == 0 && (blk == 0 || blk == 100);                                                    \
    if (mtr) { mt[0] = wall_clock64(); mslot = atomicAdd(g_vox_trace, 1ull); }
#define MLP_TR(k) if (mtr) mt[k] = wall_clock64();
#define MLP_TR_END if (mtr && mslot < VOX_TRACE_CAP) { unsigned long long* r = g_vox_trace + 16 * (mslot + 1); r[0] = blk == 0 ? 4 : 5; r[1] = ATTN; r[2] = 0; \
        for (int i_ = 0; i_ < 8; ++i_) r[3 + i_] = mt[i_]; r[11] = wall_clock64(); }
#else
#define MLP_TR_DECL
#define MLP_TR(k)
#define MLP_TR_END
#endif
// ATTN = 0: no attention in the launch; 8 / 16: its chunk count (<= 256 / <= 512 visible tokens: one / two 32-token chunks per wave)
// MULTI: every layer of the stack in this launch (ATTN form only)
template <int ATTN, bool MULTI = false>
__global__ __launch_bounds__(512) void k_talker_mlp(TalkerMlpArgs a) {
    static_assert(!MULTI || ATTN != 0, "MULTI: the attention runs in the launch");
    // MULTI: register arrays that only some waves fill are zeroed at their declaration.  Left undefined inside the layer loop they are
    // not promoted to registers at all (every weight row stored to scratch memory behind a wait of its own: 908 bytes per lane); an
    // empty asm definition instead of the zeroes makes the compiler wait for whatever is in flight at that point (behind stage O's
    // publish: the write-through store of x').  The zeroing folds into the branch that does not load.
    auto fresh = [](auto& arr) {
        if constexpr (MULTI) {
            uint4* p = reinterpret_cast<uint4*>(&arr);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(arr) / sizeof(uint4)); ++i) p[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    using TalkerAttnSmem = TalkerAttnSmemT<ATTN>;
    constexpr int H = 2048, F = 6144;
    // x' (bf16 row) | h | the attention row; ATTN: carved from the attention's LDS (dead by then — barrier below)
    __shared__ __attribute__((aligned(16))) unsigned char smem[ATTN ? sizeof(TalkerAttnSmem) : (H + F) * 2];
    static_assert(!ATTN || sizeof(TalkerAttnSmem) >= (H + F + H) * 2, "LDS carve");
    uint4* const xb = reinterpret_cast<uint4*>(smem);
    uint4* const hb = xb + H / 8;
    uint4* const ab = hb + F / 8;
    const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), blk = blockIdx.x;
    int tid = tid0, lane = tid0 & 63;
    // The launch's state words {epoch, error, poll bound, test hook}: read by ONE thread per block and handed to the others through
    // LDS.  (Every thread reading them with the L1-bypassing loads such words need put 6144 wave requests per launch on one L2 line:
    // with loads returning in order, every wave's first real operand waited behind that queue — the attention's page ids arrived 8.7 us
    // after entry instead of 3.2, tools/attn_in_layer_stamps.py.)
    __shared__ unsigned wsh[4];
    unsigned w_ep = 0, w_spins = 0, w_inj = 0;
    if (ATTN && tid == 0) {
        w_ep = __hip_atomic_load(a.epoch, VOX_RLX_AGENT);
        w_spins = __hip_atomic_load(a.epoch + 2, VOX_RLX_AGENT);
        w_inj = __hip_atomic_load(a.epoch + 3, VOX_RLX_AGENT);
    }
    const bool attn_blk = ATTN && blk < 16;
    const int nl = MULTI ? a.n_layers : 1;
    unsigned ep = 0, max_spins = 0;
    bool drop_first = false;
    __shared__ unsigned xcar[4];      // MULTI: the block's four pairs of x'' (stage D, waves 4..7) are the residual pairs of the next layer's stage O (waves 0..3)
    int l = 0;
    do {
    MLP_TR_DECL      // (development builds: one record per layer)
    if constexpr (MULTI) {
        // (the thread index made opaque per layer: left visible, every per-thread address of the body — granule pointers of four gathers,
        // weight row offsets — is hoisted out of the layer loop and kept in registers across it: 908 bytes of scratch per lane)
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
    }
    // the layer's operands: the launch arguments, or (MULTI) row l of the layer table through scalar loads
    const bf16_t *p_wo = a.wo, *p_wgate = a.wgate, *p_wup = a.wup, *p_wdown = a.wdown, *p_ln2 = a.ln2, *p_wqkv_next = a.wqkv_next, *p_ln1_next = a.ln1_next;
    AttnArgs at = a.at;
    if constexpr (MULTI) {
        static_assert(sizeof(TalkerLayerW) == 9 * sizeof(void*), "TalkerLayerW: nine pointers");
        const VOX_CONST_AS unsigned long long* lp = (const VOX_CONST_AS unsigned long long*)(a.tab + l);
        p_wo = assume_global(reinterpret_cast<const bf16_t*>(lp[0])); p_wgate = assume_global(reinterpret_cast<const bf16_t*>(lp[1])); p_wup = assume_global(reinterpret_cast<const bf16_t*>(lp[2]));
        p_wdown = assume_global(reinterpret_cast<const bf16_t*>(lp[3])); p_ln2 = assume_global(reinterpret_cast<const bf16_t*>(lp[4]));
        at.qn = assume_global(reinterpret_cast<const bf16_t*>(lp[7])); at.kn = assume_global(reinterpret_cast<const bf16_t*>(lp[8]));
        p_wqkv_next = l + 1 < nl ? assume_global(reinterpret_cast<const bf16_t*>(lp[9 + 5])) : nullptr;
        p_ln1_next = l + 1 < nl ? assume_global(reinterpret_cast<const bf16_t*>(lp[9 + 6])) : nullptr;
        at.kv = a.at.kv + (size_t)l * a.kv_layer_stride;
        at.kv_w = a.at.kv_w + (size_t)l * a.kv_layer_stride;
    }
    auto park_words = [&]() { if (tid == 0) { wsh[0] = w_ep; wsh[1] = (w_ep + (unsigned)l) * 64u + 4u; wsh[2] = w_spins; wsh[3] = w_inj; } };
    // The other 240 blocks hold their opening burst back a little (a.burst_delay x ~0.4 us): 41 MB of weight requests issued at launch
    // queue in front of the attention's K/V tiles (which only leave their blocks after two dependent round trips: arguments, page ids)
    // and the attention — the critical path of the layer — finished at 15 us instead of 9 (tools/mlp_trace.py).
    if (ATTN && !attn_blk)
        for (int d = 0, nd = l == 0 ? (a.burst_delay & 255) : (a.burst_delay >> 8); d < nd; ++d) __builtin_amdgcn_s_sleep(16);
    // ---- stage O's weight rows (waves 0..3).  The attention blocks request them when their K/V tiles are parked in LDS: nothing of
    // the attention waits behind them (loads return in order), and they land during its arithmetic
    uint4 wo[2][4];
    fresh(wo);
    unsigned resw = 0;
    auto load_o = [&]() {
        if (wave < 4) {
            const int n0 = 2 * (blk * 4 + wave);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint4* wr = reinterpret_cast<const uint4*>(p_wo + (size_t)(n0 + r) * H);
#pragma unroll
                for (int j = 0; j < 4; ++j) wo[r][j] = ldg_nt(wr + lane + 64 * j);
            }
            if (ATTN) resw = MULTI && l > 0 ? xcar[wave] : reinterpret_cast<const unsigned*>(a.x_in)[blk * 4 + wave];
        }
    };
    if (!attn_blk) load_o();
    // ---- weights of stage C, first pair of every wave: requested at once (waves 4..7 have nothing else to do until x' arrives); the
    // attention blocks request them when the attention is done (its tiles need the registers) — they land during the hand-off
    const int p1 = blk * 12 + wave;                                     // 3072 pairs: 12 per block
    uint4 wg1[2][4], wu1[2][4];
    auto load_c1 = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint4* gr = reinterpret_cast<const uint4*>(p_wgate + (size_t)(2 * p1 + r) * H);
            const uint4* ur = reinterpret_cast<const uint4*>(p_wup + (size_t)(2 * p1 + r) * H);
#pragma unroll
            for (int j = 0; j < 4; ++j) { wg1[r][j] = ldg_nt(gr + lane + 64 * j); wu1[r][j] = ldg_nt(ur + lane + 64 * j); }
        }
    };
    if (!attn_blk) load_c1();
    if (attn_blk) {
        // q head blk: kv head blk / 2, half blk % 2 of its two-head group (the launch form's head split); row 0
        TalkerAttnSmem& sm = *reinterpret_cast<TalkerAttnSmem*>(smem);
        // MULTI: q head blk, k / v head blk / 2 of the new token -> sm.Raw: layer 0 from the plain row of the projection launch in front,
        // later layers from the previous layer's stage-A granules (192 of the 2048: one per thread of waves 0..2)
        auto stage_raw = [&]() {
            if constexpr (MULTI) {
                if (tid < 192) {
                    const int part = tid >> 6, t = tid & 63;
                    const int gi = part == 0 ? blk * 64 + t : (part == 1 ? 1024 : 1536) + (blk >> 1) * 64 + t;
                    unsigned v = 0;
                    if (l == 0) {
                        v = reinterpret_cast<const unsigned*>(a.at.qkv)[gi];
                    } else {
                        const unsigned want = (wsh[0] + (unsigned)l - 1u) * 64u + 5u, bound = wsh[2];
                        for (unsigned spin = 0;; ++spin) {
                            const unsigned long long x = __hip_atomic_load(a.gq + gi, VOX_RLX_AGENT);
                            if ((unsigned)(x >> 32) == want) { v = (unsigned)x; break; }
                            if (spin > bound) { atomicCAS(a.err, 0u, 0x1500u); break; }
                            if ((spin & 63u) == 63u && __hip_atomic_load(a.err, VOX_RLX_AGENT) != 0u) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    reinterpret_cast<unsigned*>(sm.Raw)[tid] = v;
                }
                __syncthreads();
            }
        };
        attn_decode8_block<128, 1, 8, (ATTN ? ATTN : 8), 512, MULTI>(at, sm, 2, blk >> 1, blk & 1, 0, a.gattn, &wsh[1], false,
                                              [&]() { park_words(); load_o(); }, a.epoch, stage_raw);
        load_c1();
    } else if (ATTN && l == 0) {
        park_words();
    }
    uint4 av[4];
    fresh(av);
    if (ATTN) __syncthreads();                         // state words parked; (attention blocks) the attention's LDS is dead: xb / hb / ab may be written
    if (!ATTN && wave < 4) {       // stage O's operand row (a plain row of the attention launch in front): requested before anything below waits
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = reinterpret_cast<const uint4*>(a.attn)[lane + 64 * j];
        // (the wave's residual pair: a uniform address, i.e. a scalar load with a wait of its own — behind the opening requests, not between them)
        asm volatile("" ::: "memory");
        resw = reinterpret_cast<const unsigned*>(a.x_in)[blk * 4 + wave];
    }
    if (!ATTN) {
        // (without the attention in front: plain loads of a uniform address — scalar loads; the words were written before this launch started
        // and a kernel boundary stands in between — BEHIND the opening weight requests: as L1-bypassing vector loads at the top of the kernel
        // they were a round trip in front of every wave's first weight request)
        asm volatile("" ::: "memory");
        const VOX_CONST_AS unsigned* wp = (const VOX_CONST_AS unsigned*)a.epoch;      // (constant address space: scalar loads)
        w_ep = wp[0]; w_spins = wp[2]; w_inj = wp[3];
    }
    if (l == 0) {
        ep = ATTN ? wsh[0] : w_ep; max_spins = ATTN ? wsh[2] : w_spins;
        drop_first = blk == 1 && (ATTN ? wsh[3] : w_inj) != 0u;
    }
    const unsigned tag0 = (ep + (unsigned)l) * 64u;
    MLP_TR(1)
    // ---------------- stage O: x' = x + Wo . attn ----------------
    if (ATTN) {
        // The attention row: 1024 granules from blocks 0..15.  The wait is long (the whole attention, ~10 us) and the same for every
        // block: ONE wave per block watches one sentinel granule with s_sleep between its polls (240 blocks sweeping 1024 granules each
        // for the whole time take the memory side away from the attention's K/V tiles and from the weight stream: guide, polling-cost);
        // the others wait at the barrier.  Then the gather proper, by every thread.
        if (wave == 7 && !attn_blk) {
            for (unsigned spin = 0; spin <= max_spins; ++spin) {
                if ((unsigned)(__hip_atomic_load(a.gattn + 1023, VOX_RLX_AGENT) >> 32) == tag0 + 4u) break;
                for (int q = 0; q < a.poll_sleep; ++q) __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        gran_gather_lds_all<1024>(a.gattn, tag0 + 4u, reinterpret_cast<unsigned*>(ab), tid, a.err, 0x1400u, max_spins, a.poll_delay >> 24);
        __syncthreads();
    }
    MLP_TR(2)
    // U: 24 chunks whose content depends on the wave's role — waves 0..3: the second gate / up pair (gate rows in U[0..7], up rows in
    // U[8..15]); waves 4..7: their two down-projection rows (U[0..23]).  One array: as separate ones the compiler adds the roles' registers up.
    uint4 U[24];
    fresh(U);
    if (wave < 4) {
        const int pr = blk * 4 + wave;
        if (ATTN) {
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = ab[lane + 64 * j];
        }
        float acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float d = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) d = dot8(wo[r][j], av[j], d);
            acc[r] = butterfly<64>(d);
        }
        if (lane == 0) {
            const bf16_t r0 = f2bf(bflo(resw) + bf2f(f2bf(acc[0]))), r1 = f2bf(bfhi(resw) + bf2f(f2bf(acc[1])));
            if (!(drop_first && l == 0)) gran_write(a.gx + pr, tag0 + 1u, r0, r1);
        }
    }
    // ---------------- stage C: h = silu(Wg . n) * (Wu . n), n = rmsnorm(x', ln2) ----------------
    uint4 nwv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) nwv[j] = ldg(reinterpret_cast<const uint4*>(p_ln2) + lane + 64 * j);
    const int p2 = blk * 12 + 8 + wave;                                 // second pair: waves 0..3
    auto load_c2 = [&]() {
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint4* gr = reinterpret_cast<const uint4*>(p_wgate + (size_t)(2 * p2 + r) * H);
                const uint4* ur = reinterpret_cast<const uint4*>(p_wup + (size_t)(2 * p2 + r) * H);
#pragma unroll
                for (int j = 0; j < 4; ++j) { U[r * 4 + j] = ldg_nt(gr + lane + 64 * j); U[8 + r * 4 + j] = ldg_nt(ur + lane + 64 * j); }
            }
        }
    };
    // VOX_MLP_C2_LATE=1 (measured, no gain: 2.485 vs 2.464 ms per frame): the second pair's rows requested AFTER x' has been
    // gathered, so that the gather's polls do not come back behind 64 KB of HBM loads per block.
    if (!VOX_MLP_C2_LATE) load_c2();
    __syncthreads();                                   // (parks waves 4..7 until waves 0..3 have published their x' pairs)
    MLP_TR(3)
    gran_gather_lds_all<1024>(a.gx, tag0 + 1u, reinterpret_cast<unsigned*>(xb), tid, a.err, 0x1100u, max_spins, a.poll_delay & 255u);
    if (VOX_MLP_C2_LATE) load_c2();
    __syncthreads();
    MLP_TR(4)
    // Stage D's rows belong to waves 4..7 and are requested HERE, a whole stage ahead.  A block's requests are served in the order it made
    // them (its own queue to the memory side, ~11 bytes per clock per CU when every CU streams): requested when stage C was done (round 5:
    // by waves 0..3, which had no registers for them earlier) the 96 KB stood in front of the h gather's polls and came back 5 us later —
    // h was gathered 8.5 us after x' (tools/mlp_trace.py).  From here they travel under stage C's arithmetic.
    auto load_d = [&]() {
        if (wave >= 4) {
            const int n0 = 2 * (blk * 4 + wave - 4);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint4* wr = reinterpret_cast<const uint4*>(p_wdown + (size_t)(n0 + r) * F);
#pragma unroll
                for (int j = 0; j < 12; ++j) U[r * 12 + j] = ldg_nt(wr + lane + 64 * j);
            }
        }
    };
    if (VOX_MLP_D_EARLY) load_d();
    uint4 xv[4];
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xb[lane + 64 * j];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = sq8(xv[j], s);
        s = butterfly<64>(s);
        const float rinv = 1.0f / sqrtf(s / (float)H + a.eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = norm_chunk(xv[j], nwv[j], rinv);
    }
    auto ffn_pair = [&](const uint4* wg, const uint4* wu, int pr) {      // (two gate rows, two up rows: four chunks each)
        bf16_t hv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float dg = 0.0f, du = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) dg = dot8(wg[r * 4 + j], xv[j], dg);
#pragma unroll
            for (int j = 0; j < 4; ++j) du = dot8(wu[r * 4 + j], xv[j], du);
            const float gg = bfround(butterfly<64>(dg)), uu = bfround(butterfly<64>(du));
            hv[r] = f2bf(bfround(silu_c(gg)) * uu);
        }
        if (lane == 0) gran_write(a.gh + pr, tag0 + 2u, hv[0], hv[1]);
    };
    ffn_pair(&wg1[0][0], &wu1[0][0], p1);
    if (wave < 4) ffn_pair(U, U + 8, p2);
    // ---------------- stage D: x'' = x' + Wd . h ----------------
    {
        const int pr = blk * 4 + wave - 4;
        if (!VOX_MLP_D_EARLY) load_d();
        __syncthreads();                               // (parks waves 4..7 while waves 0..3 finish their second pair)
        MLP_TR(5)
        gran_gather_lds_all<3072>(a.gh, tag0 + 2u, reinterpret_cast<unsigned*>(hb), tid, a.err, 0x1200u, max_spins, (a.poll_delay >> 8) & 255u);
        __syncthreads();
        MLP_TR(6)
        if (wave >= 4) {
            const unsigned resw = reinterpret_cast<const unsigned*>(xb)[pr];
            float acc[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float d = 0.0f;
#pragma unroll
                for (int j = 0; j < 12; ++j) d = dot8(U[r * 12 + j], hb[lane + 64 * j], d);
                acc[r] = butterfly<64>(d);
            }
            const bf16_t r0 = f2bf(bflo(resw) + bf2f(f2bf(acc[0]))), r1 = f2bf(bfhi(resw) + bf2f(f2bf(acc[1])));
            if (lane == 0) {
                reinterpret_cast<unsigned*>(a.x_out)[pr] = (unsigned)r0 | ((unsigned)r1 << 16);      // (the next layer's residual reads it)
                if (p_wqkv_next) gran_write(a.gx + pr, tag0 + 3u, r0, r1);
            }
            if (MULTI && lane == 0) xcar[pr & 3] = (unsigned)r0 | ((unsigned)r1 << 16);      // (the next layer's stage O, wave pr & 3, adds its product to it)
        }
    }
    // ---------------- stage A of the next layer: qkv = Wqkv . rmsnorm(x'', ln1)  (2048 pairs: one per wave) ----------------
    if (p_wqkv_next) {
        const int pr = blk * 8 + wave, n0 = 2 * pr;
        uint4 wq[2][4], nw1[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint4* wr = reinterpret_cast<const uint4*>(p_wqkv_next + (size_t)(n0 + r) * H);
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[r][j] = ldg_nt(wr + lane + 64 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) nw1[j] = ldg(reinterpret_cast<const uint4*>(p_ln1_next) + lane + 64 * j);
        __syncthreads();                               // (parks waves 4..7 while waves 0..3 finish the down projection; xb is free)
        gran_gather_lds_all<1024>(a.gx, tag0 + 3u, reinterpret_cast<unsigned*>(xb), tid, a.err, 0x1300u, max_spins, (a.poll_delay >> 16) & 255u);
        __syncthreads();
        MLP_TR(7)
        uint4 yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = xb[lane + 64 * j];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = sq8(yv[j], s);
        s = butterfly<64>(s);
        const float rinv = 1.0f / sqrtf(s / (float)H + a.eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = norm_chunk(yv[j], nw1[j], rinv);
        float acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float d = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) d = dot8(wq[r][j], yv[j], d);
            acc[r] = butterfly<64>(d);
        }
        if (lane == 0) {
            if (MULTI) gran_write(a.gq + pr, tag0 + 5u, f2bf(acc[0]), f2bf(acc[1]));      // the next layer's attention blocks gather it
            else reinterpret_cast<unsigned*>(a.qkv_out)[pr] = (unsigned)f2bf(acc[0]) | ((unsigned)f2bf(acc[1]) << 16);
        }
    }
    if (MULTI) __syncthreads();      // stage A's reads of xb are done: the next layer's tiles / gathers may overwrite the buffers
    MLP_TR_END
    } while (MULTI && ++l < nl);      // layers
    if (blk == 0 && tid == 0) __hip_atomic_store(a.epoch, ep + (unsigned)nl, VOX_RLX_AGENT);
    if (drop_first && tid == 0) atomicSub(a.epoch + 3, 1u);
}

bool vox_talker_mlp_supported(const TalkerMlpCall& c) { return c.hidden == 2048 && c.nq == 2048 && c.ffn == 6144; }
// the decode attention inside the launch: one row, 16 q heads of 128 on 8 kv heads, <= 256 visible tokens (8 chunks), fused decode mode
static void fill_attn_args(AttnArgs& a, const AttnCall& c);
bool vox_talker_attn_supported(const AttnCall& c) {
    // (<= 256 visible tokens: one chunk per wave.  Two chunks per wave up to 512 — k_talker_mlp<16>, kept compiled and tested — measured
    // SLOWER than the attention launches there: 2.26 vs 2.17 ms per frame at 325 tokens, 2.26 vs 2.20 at 480; VOX_TALKER_ATTN512=1 for A/B)
    static const bool a512 = [] { const char* e = getenv("VOX_TALKER_ATTN512"); return e && e[0] == '1'; }();
    return c.qkv && c.Nq == 1 && c.D == 128 && c.Hq == 16 && c.Hkv == 8 && c.max_kvlen > VOX_TC && c.max_kvlen <= (a512 ? 16 : 8) * VOX_TC;
}
// First-pass hold-back of the persistent kernels' gathers (gran_poll_all), one byte per gather site in units of 128 clocks.  The defaults
// are the round-6 sweep's (tools/poll_delay_sweep.sh, profiles/round6_poll_delay_sweep.txt); the environment overrides them for A/B runs
// (hex or decimal, read when a launch is first built — frame graphs captured afterwards keep the value).
#ifndef VOX_DS_POLL_DELAY_DEFAULT
#define VOX_DS_POLL_DELAY_DEFAULT 0x18100604u
#endif
#ifndef VOX_MLP_POLL_DELAY_DEFAULT
#define VOX_MLP_POLL_DELAY_DEFAULT 0x04021004u
#endif
static unsigned vox_poll_delay(const char* name, unsigned dflt) {
    const char* e = getenv(name);
    return e ? (unsigned)strtoul(e, nullptr, 0) : dflt;
}
int vox_launch_talker_mlp(hipStream_t st, const TalkerMlpCall& c) {
    if (!vox_talker_mlp_supported(c)) return vox_fail(VOX_ERR_INVALID, "talker_mlp: unsupported shape");
    TalkerMlpArgs a{};
    a.wo = (const bf16_t*)c.wo; a.wgate = (const bf16_t*)c.wgate; a.wup = (const bf16_t*)c.wup; a.wdown = (const bf16_t*)c.wdown;
    a.ln2 = (const bf16_t*)c.ln2; a.attn = (const bf16_t*)c.attn; a.x_in = (const bf16_t*)c.x; a.x_out = (bf16_t*)c.x;
    a.gx = (unsigned long long*)c.gran; a.gh = a.gx + 1024; a.epoch = c.epoch; a.err = c.err; a.eps = c.eps;
    a.wqkv_next = (const bf16_t*)c.wqkv_next; a.ln1_next = (const bf16_t*)c.ln1_next; a.qkv_out = (bf16_t*)c.qkv_out;
    if (c.wqkv_next && (c.nqkv != 4096 || !c.ln1_next || !c.qkv_out)) return vox_fail(VOX_ERR_INVALID, "talker_mlp: bad next-layer qkv");
    a.poll_delay = vox_poll_delay("VOX_MLP_POLL_DELAY", VOX_MLP_POLL_DELAY_DEFAULT);
    if (c.attn_call) {
        const AttnCall& ac = *c.attn_call;
        if (!vox_talker_attn_supported(ac)) return vox_fail(VOX_ERR_INVALID, "talker_mlp: attention shape not supported in the launch");
        fill_attn_args(a.at, ac);
        static const int hoist_on = [] { const char* e = getenv("VOX_ATTN_HOIST"); return !(e && e[0] == '0'); }();
        a.at.hoist = hoist_on;
        a.gattn = a.gx + 4096;
        const int delay = [] { const char* e = getenv("VOX_TALKER_ATTN_DELAY"); return e ? atoi(e) : 8; }();       // (read per launch built: graphs keep it)
        a.burst_delay = delay < 0 ? 0 : (delay > 64 ? 64 : delay);      // ([7:0] the launch's first layer, [15:8] later layers of the all-layer form)
        const int psl = [] { const char* e = getenv("VOX_TALKER_ATTN_POLL"); return e ? atoi(e) : 1; }();
        a.poll_sleep = psl < 1 ? 1 : (psl > 64 ? 64 : psl);
        if (c.layer_tab) {
            // every layer in ONE launch (<= 256 visible tokens); granules: x' 1024 | h 3072 | attention row 1024 | q k v 2048
            if (c.n_layers < 1) return vox_fail(VOX_ERR_INVALID, "talker_mlp: all-layer form without layers");
            if (!ac.qn || !ac.kn || !ac.cs) return vox_fail(VOX_ERR_INVALID, "talker_mlp: all-layer form needs head norms and a RoPE table");
            a.tab = (const TalkerLayerW*)c.layer_tab; a.n_layers = c.n_layers; a.kv_layer_stride = c.kv_layer_stride;
            a.gq = a.gx + 5120;
            const int d2 = [] { const char* e = getenv("VOX_TALKER_MULTI_DELAY"); return e ? atoi(e) : 8; }();
            a.burst_delay |= (d2 < 0 ? 0 : (d2 > 64 ? 64 : d2)) << 8;
            if (ac.max_kvlen <= 8 * VOX_TC) hipLaunchKernelGGL((k_talker_mlp<8, true>), dim3(256), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_talker_mlp<16, true>), dim3(256), dim3(512), 0, st, a);
            return VOX_OK;
        }
        if (ac.max_kvlen <= 8 * VOX_TC) hipLaunchKernelGGL(k_talker_mlp<8>, dim3(256), dim3(512), 0, st, a);
        else hipLaunchKernelGGL(k_talker_mlp<16>, dim3(256), dim3(512), 0, st, a);
        return VOX_OK;
    }
    hipLaunchKernelGGL(k_talker_mlp<0>, dim3(256), dim3(512), 0, st, a);
    return VOX_OK;
}

bool vox_depth_step_supported(const DepthStepCall& c) {
    return c.hidden == 1024 && c.heads == 16 && c.kv_heads == 8 && c.head_dim == 128 && c.ffn == 3072 && c.vocab == 2048 && c.qk_norm &&
           !c.qkv_bias && c.rope_dim == 128 && !c.rope_interleave && c.n_tokens >= 2 && c.n_tokens <= 16 && c.n_layers >= 1 && c.page_size >= c.n_tokens;
}
int vox_launch_depth_step(hipStream_t st, const DepthStepCall& c) {
    if (!vox_depth_step_supported(c)) return vox_fail(VOX_ERR_INVALID, "depth_step: unsupported shape");
    DepthStepArgs a{};
    a.layers = (const DepthLayerW*)c.layers_dev; a.n_layers = c.n_layers; a.final_norm = (const bf16_t*)c.final_norm;
    a.head_w = (const bf16_t*)c.head_w; a.x_in = (const bf16_t*)c.x_in; a.logits = (bf16_t*)c.logits;
    a.gx = (unsigned long long*)c.gran; a.gqkv = a.gx + 512; a.gh = a.gqkv + 2048;
    a.epoch = c.epoch; a.err = c.err; a.kv_layer_stride = c.kv_layer_stride; a.eps = c.eps;
    a.poll_delay = vox_poll_delay("VOX_DS_POLL_DELAY", VOX_DS_POLL_DELAY_DEFAULT);
    if (c.pick_logits) {
        if (!c.pick_tab || !c.pick_emb || !c.pick_out || !c.pick_feat || c.pick_vocab <= 0 || c.pick_vocab > 65536 || c.pick_vocab % 4 || c.pick_H % 8)
            return vox_fail(VOX_ERR_INVALID, "depth_step: bad fused-pick arguments");
        a.pick_logits = (const bf16_t*)c.pick_logits; a.pick_tab = (const bf16_t*)c.pick_tab; a.pick_emb = (const bf16_t*)c.pick_emb;
        a.pick_out = c.pick_out; a.pick_feat = (bf16_t*)c.pick_feat; a.pick_vocab = c.pick_vocab; a.pick_H = c.pick_H; a.pick_init = c.pick_init;
        if (c.pick_top_k > 0) {
            if (c.pick_vocab != 2048 || c.pick_top_k > 256 || !(c.pick_temperature > 0.0f) || c.pick_min_p > 1.0f)
                return vox_fail(VOX_ERR_INVALID, "depth_step: the sampled pick takes a 2048-entry vocabulary, top_k <= 256, temperature > 0");
            a.pick_top_k = c.pick_top_k; a.pick_top_p = c.pick_top_p; a.pick_min_p = c.pick_min_p; a.pick_temperature = c.pick_temperature;
            a.pick_seed = c.pick_seed; a.pick_offset = c.pick_offset; a.pick_offset_mul = c.pick_offset_mul; a.pick_offset_dev = c.pick_offset_dev;
        }
    }
    AttnArgs& at = a.at;
    at.kv = (const bf16_t*)c.kv; at.kv_w = (bf16_t*)c.kv; at.cs = c.cs; at.eps = c.eps; at.scale = c.scale;
    at.Hq = c.heads; at.Hkv = c.kv_heads; set_page_size(at, c.page_size); at.table_max_pos = c.table_max_pos;
    at.rot = c.rope_dim; at.interleave = 0; at.fixed_kvlen = c.n_tokens; at.fixed_pos = c.n_tokens - 1; at.identity_pages = 1;
    at.out = nullptr; at.out_frag = nullptr;
    switch (c.n_tokens) {
#define VOX_DS(NT_) case NT_: hipLaunchKernelGGL(k_depth_step<NT_>, dim3(256), dim3(512), 0, st, a); return VOX_OK;
        VOX_DS(2) VOX_DS(3) VOX_DS(4) VOX_DS(5) VOX_DS(6) VOX_DS(7) VOX_DS(8) VOX_DS(9) VOX_DS(10) VOX_DS(11) VOX_DS(12)
        VOX_DS(13) VOX_DS(14) VOX_DS(15) VOX_DS(16)
#undef VOX_DS
        default: break;
    }
    return vox_fail(VOX_ERR_INVALID, "depth_step: n_tokens out of range");
}

static void fill_attn_args(AttnArgs& a, const AttnCall& c) {
    a.q = (const bf16_t*)c.q; a.kv = (const bf16_t*)c.kv; a.q_req = c.q_req; a.q_kvlen = c.q_kvlen;
    a.indptr = c.indptr; a.indices = c.indices; a.part_o = c.part_o; a.part_ml = c.part_ml; a.scale = c.scale;
    a.Hq = c.Hq; a.Hkv = c.Hkv; set_page_size(a, c.page_size); a.max_chunks = c.max_chunks;
    a.qkv = (const bf16_t*)c.qkv; a.kv_w = (bf16_t*)const_cast<void*>(c.kv); a.qn = (const bf16_t*)c.qn;
    a.kn = (const bf16_t*)c.kn; a.cs = c.cs; a.pos = c.pos; a.page = c.page; a.slot = c.slot; a.eps = c.eps;
    a.rot = c.rot; a.interleave = c.interleave; a.table_max_pos = c.table_max_pos;
    a.ptab = c.ptab; a.pt_stride = c.pt_stride; a.fixed_kvlen = c.fixed_kvlen; a.fixed_pos = c.fixed_pos;
    a.identity_pages = c.identity_pages;
    a.out = nullptr;
}

// which compile-time token count a short-attention launch may take: NT > 0 assumes identity pages, a fixed position and a page that
// holds all NT tokens (attn_short_prefetch); anything else reads the plan arrays (NT = 0)
static inline int attn_short_nt(const AttnArgs& at) {
    return (at.identity_pages && at.fixed_pos >= 0 && at.fixed_kvlen >= 2 && at.fixed_kvlen <= 32 && at.page_size >= at.fixed_kvlen) ? at.fixed_kvlen : 0;
}
// short-context decode attention: D 128, two q heads per kv head, full-width NeoX RoPE, <= 16 visible tokens
bool vox_attn_short_supported(const AttnCall& c) {
    // <= 16 visible tokens: any plan; 17 .. 32: the compile-time form only (identity pages, fixed length and position: the depth loops)
    const bool len_ok = c.max_kvlen <= 16 || (c.max_kvlen <= 32 && c.identity_pages && c.fixed_pos >= 0 && c.fixed_kvlen == c.max_kvlen && c.page_size >= c.fixed_kvlen);
    return c.qkv && c.D == 128 && len_ok && c.Nq >= 1 && c.Hkv > 0 && (c.Hq == 2 * c.Hkv || c.Hq == 4 * c.Hkv) && c.rot == 128 &&
           !c.interleave && c.cs;
}
int vox_launch_attn_short(hipStream_t st, const AttnCall& c) {
    if (!vox_attn_short_supported(c) || !c.out) return vox_fail(VOX_ERR_INVALID, "attn_short: unsupported shape");
    AttnArgs at{};
    fill_attn_args(at, c);
    at.out = (bf16_t*)c.out;
    at.out_frag = (bf16_t*)c.out_frag;
    const int n_pairs = c.Nq * c.Hkv * (c.Hq / (2 * c.Hkv));
    // NT > 0 = the depth-loop form: identity pages, fixed position, the page holds every visible token (attn_short_prefetch)
    switch (attn_short_nt(at)) {      // depth loop: the visible length is part of the captured graph
#define VOX_AS(NT_) case NT_: hipLaunchKernelGGL(k_attn_short<NT_>, dim3((n_pairs + 3) / 4), dim3(256), 0, st, at, n_pairs); return VOX_OK;
        VOX_AS(2) VOX_AS(3) VOX_AS(4) VOX_AS(5) VOX_AS(6) VOX_AS(7) VOX_AS(8) VOX_AS(9) VOX_AS(10) VOX_AS(11) VOX_AS(12)
        VOX_AS(13) VOX_AS(14) VOX_AS(15) VOX_AS(16) VOX_AS(17) VOX_AS(18) VOX_AS(19) VOX_AS(20) VOX_AS(21) VOX_AS(22) VOX_AS(23) VOX_AS(24)
        VOX_AS(25) VOX_AS(26) VOX_AS(27) VOX_AS(28) VOX_AS(29) VOX_AS(30) VOX_AS(31) VOX_AS(32)
#undef VOX_AS
        default: break;
    }
    if (c.max_kvlen > 16) return vox_fail(VOX_ERR_INVALID, "attn_short: 17..32 tokens need the fixed-length form");
    hipLaunchKernelGGL(k_attn_short<0>, dim3((n_pairs + 3) / 4), dim3(256), 0, st, at, n_pairs);
    return VOX_OK;
}
// true when the fused short-attention + o_proj kernel covers this call (else the caller launches the two kernels)
bool vox_attn1_linear_supported(const AttnCall& c, const LinearCall& l) {
    // one row only: with two rows each wave runs two attention passes back to back — the separate one-wave-per-(row, head)
    // kernel + a 2-row GEMV is faster there (B=2 frame 4.30 -> 4.05 ms)
    return vox_attn_short_supported(c) && c.max_kvlen <= 16 && c.Hq == 2 * c.Hkv && c.Nq == 1 && l.K == c.Hq * c.D && l.K == 2048 && l.pro == PRO_COPY &&
           l.epi == EPI_STORE && !l.x_rows && l.B == c.Nq;
}
int vox_launch_attn1_linear(hipStream_t st, const AttnCall& c, const LinearCall& l) {
    if (!vox_attn1_linear_supported(c, l)) return vox_fail(VOX_ERR_INVALID, "attn1_linear: unsupported shape");
    AttnArgs at{};
    fill_attn_args(at, c);
    LinArgs a{};
    a.W = (const bf16_t*)l.W; a.bias = (const bf16_t*)l.bias; a.residual = (const bf16_t*)l.residual; a.y = (bf16_t*)l.y;
    a.B = l.B; a.N = l.N; a.K = l.K;
    const dim3 grid((l.N + 7) / 8);
    switch (attn_short_nt(at)) {      // depth loop: the visible length is part of the captured graph
#define VOX_A1(NT_) case NT_: hipLaunchKernelGGL((k_attn1_linear<1, 4, 1, NT_>), grid, dim3(512), 0, st, at, a); return VOX_OK;
        VOX_A1(2) VOX_A1(3) VOX_A1(4) VOX_A1(5) VOX_A1(6) VOX_A1(7) VOX_A1(8) VOX_A1(9) VOX_A1(10) VOX_A1(11) VOX_A1(12)
        VOX_A1(13) VOX_A1(14) VOX_A1(15) VOX_A1(16)
#undef VOX_A1
        default: break;
    }
    hipLaunchKernelGGL((k_attn1_linear<1, 4, 1, 0>), grid, dim3(512), 0, st, at, a);
    return VOX_OK;
}
// merge partials -> bf16 out [Nq,Hq,D] (standalone op path; the engine merges inside the o_proj prologue)
__global__ __launch_bounds__(256) void k_attn_merge(const float* part_o, const float* part_ml, const int* kvlen,
                                                    bf16_t* out, int Hq, int D, int max_chunks, int total, bf16_t* out_frag) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int HD = Hq * D;
    const int row = e / HD, h = (e % HD) / D, d = e % D;
    const int nc = (kvlen[row] + VOX_TC - 1) / VOX_TC;
    const float2* ml = reinterpret_cast<const float2*>(part_ml + ((size_t)row * Hq + h) * max_chunks * 2);
    const float* po = part_o + ((size_t)row * Hq + h) * max_chunks * D + d;
    // all operands of up to 8 chunks are requested before the first is used (the max pass needs every (m, l) anyway)
    float M = -INFINITY, L = 0.0f, O = 0.0f;
    float2 mv[8];
    float ov[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mv[j] = j < nc ? ml[j] : make_float2(-INFINITY, 0.0f);
        ov[j] = j < nc ? po[(size_t)j * D] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) M = fmaxf(M, mv[j].x);
    for (int c = 8; c < nc; ++c) M = fmaxf(M, ml[c].x);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < nc) {
            const float w = exp2_c((mv[j].x - M) * VOX_LOG2E);
            L = __fmaf_rn(mv[j].y, w, L);
            O = __fmaf_rn(ov[j], w, O);
        }
    }
    for (int c = 8; c < nc; ++c) {
        const float w = exp2_c((ml[c].x - M) * VOX_LOG2E);
        L = __fmaf_rn(ml[c].y, w, L);
        O = __fmaf_rn(po[(size_t)c * D], w, O);
    }
    out[e] = f2bf(O / L);
    if (out_frag) out_frag[frag_off(row, e % HD, HD)] = f2bf(O / L);
}

// The same merge for long contexts (more than 8 chunks: the > 256-token buckets), one block of D threads per (row, head): the chunk
// maxima and sums are fetched once (coalesced), the chunk weights w_c = exp2((m_c - M) log2e) are evaluated once per chunk by thread c
// instead of once per output element inside a dependent loop, and thread d then runs the two fma chains over the chunks in ascending
// order with its partial-output loads issued eight at a time.  Same M, same w_c, same fma order: bit-identical to k_attn_merge
// (which at 64 chunks spent ~25 us per layer in 56 dependent iterations of two loads + one exp each).
#define VOX_MERGE_MAXC 128
__global__ __launch_bounds__(128) void k_attn_merge_row(const float* part_o, const float* part_ml, const int* kvlen, bf16_t* out, int Hq, int D,
                                                        int max_chunks, bf16_t* out_frag) {
    __shared__ float2 mls[VOX_MERGE_MAXC];
    __shared__ float ws[VOX_MERGE_MAXC];
    __shared__ float red[2];
    const int rh = blockIdx.x, row = rh / Hq, h = rh % Hq, d = threadIdx.x, lane = d & 63, wv = d >> 6;
    const int nc = (kvlen[row] + VOX_TC - 1) / VOX_TC;
    const float2* ml = reinterpret_cast<const float2*>(part_ml + (size_t)rh * max_chunks * 2);
    const float* po = part_o + (size_t)rh * max_chunks * D + d;
    float m = -INFINITY;
    for (int c = d; c < nc; c += blockDim.x) {
        const float2 v = ml[c];
        mls[c] = v;
        m = fmaxf(m, v.x);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, VOX_WAVE));
    if (lane == 0) red[wv] = m;
    __syncthreads();
    const float M = blockDim.x > 64 ? fmaxf(red[0], red[1]) : red[0];
    for (int c = d; c < nc; c += blockDim.x) ws[c] = exp2_c((mls[c].x - M) * VOX_LOG2E);
    __syncthreads();
    float L = 0.0f, O = 0.0f;
    int c = 0;
    for (; c + 8 <= nc; c += 8) {
        float ov[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = po[(size_t)(c + j) * D];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = ws[c + j];
            L = __fmaf_rn(mls[c + j].y, w, L);
            O = __fmaf_rn(ov[j], w, O);
        }
    }
    for (; c < nc; ++c) {
        const float w = ws[c];
        L = __fmaf_rn(mls[c].y, w, L);
        O = __fmaf_rn(po[(size_t)c * D], w, O);
    }
    if (d < D) {
        const bf16_t r = f2bf(O / L);
        out[(size_t)rh * D + d] = r;
        if (out_frag) out_frag[frag_off(row, h * D + d, Hq * D)] = r;
    }
}

int vox_launch_attn_merge(hipStream_t st, const float* part_o, const float* part_ml, const int* kvlen, void* out,
                          int Nq, int Hq, int D, int max_chunks, void* out_frag) {
    const int total = Nq * Hq * D;
    if (total <= 0) return VOX_OK;
    static const bool row_on = [] { const char* e = getenv("VOX_ATTN_MERGE_ROW"); return !(e && e[0] == '0'); }();
    if (row_on && max_chunks > 8 && max_chunks <= VOX_MERGE_MAXC && (D == 64 || D == 128)) {
        hipLaunchKernelGGL(k_attn_merge_row, dim3(Nq * Hq), dim3(D), 0, st, part_o, part_ml, kvlen, (bf16_t*)out, Hq, D, max_chunks,
                           (bf16_t*)out_frag);
        return VOX_OK;
    }
    hipLaunchKernelGGL(k_attn_merge, dim3((total + 255) / 256), dim3(256), 0, st, part_o, part_ml, kvlen,
                       (bf16_t*)out, Hq, D, max_chunks, total, (bf16_t*)out_frag);
    return VOX_OK;
}

// ================================================================================================
// small elementwise / gather kernels
// ================================================================================================
// dst[b*dst_stride + i] = table[ids[b*id_stride+id_off]*H + i]
__global__ __launch_bounds__(256) void k_gather_rows(const bf16_t* table, const int* ids, int id_stride, int id_off,
                                                     bf16_t* dst, long dst_stride, int H, int vocab) {
    const int b = blockIdx.y;
    int id = ids[(size_t)b * id_stride + id_off];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const uint4* src = reinterpret_cast<const uint4*>(table + (size_t)id * H);
    uint4* d = reinterpret_cast<uint4*>(dst + (size_t)b * dst_stride);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < (H >> 3); i += gridDim.x * 256) d[i] = src[i];
}
int vox_launch_gather(hipStream_t st, const void* table, const int* ids, int id_stride, int id_off, void* dst,
                      long dst_stride, int B, int H, int vocab) {
    if (B <= 0) return VOX_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3((H / 8 + 255) / 256, B), dim3(256), 0, st, (const bf16_t*)table, ids,
                       id_stride, id_off, (bf16_t*)dst, dst_stride, H, vocab);
    return VOX_OK;
}

// Qwen3 input mix (qwen3_tts.py:1836-1852): e = mask ? bf16(text + codec_emb[id0]) : text; y = bf16(e + feat)
__global__ __launch_bounds__(256) void k_qwen3_mix(const bf16_t* text, const bf16_t* codec_table, const int* ids,
                                                   int id_stride, const uint8_t* mask, const bf16_t* feat, bf16_t* y,
                                                   int H, int vocab, Qwen3Shadow sh, const uint64_t* rng) {
    const int b = blockIdx.y;
    int id = ids[(size_t)b * id_stride];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const bool m = mask[b] != 0;
    // decode frames of an engine with a status word: the row's inputs also go to the shadow slot of this frame (counter parity)
    const size_t srow = sh.rng ? (size_t)(*rng & 1) * sh.max_batch + b : 0;
    if (sh.rng && blockIdx.x == 0) {
        for (int j = threadIdx.x; j < sh.G1; j += 256) sh.ids[srow * sh.G1 + j] = ids[(size_t)b * id_stride + j];
        if (threadIdx.x == 0) sh.mask[srow] = mask[b];
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H; i += gridDim.x * 256) {
        const size_t j = (size_t)b * H + i;
        bf16_t e = text[j];
        if (m) e = f2bf(bf2f(e) + bf2f(codec_table[(size_t)id * H + i]));
        const bf16_t f = feat[j];
        if (sh.rng) sh.feat[srow * H + i] = f;
        y[j] = f2bf(bf2f(e) + bf2f(f));
    }
}
int vox_launch_qwen3_mix(hipStream_t st, const void* text, const void* codec_table, const int* ids, int id_stride,
                         const uint8_t* mask, const void* feat, void* y, int B, int H, int vocab, const Qwen3Shadow* shadow,
                         const uint64_t* rng) {
    if (B <= 0) return VOX_OK;
    Qwen3Shadow sh{};
    if (shadow && rng && shadow->rng && shadow->G1 == id_stride && shadow->H == H && B <= shadow->max_batch) sh = *shadow;
    hipLaunchKernelGGL(k_qwen3_mix, dim3((H + 255) / 256, B), dim3(256), 0, st, (const bf16_t*)text,
                       (const bf16_t*)codec_table, ids, id_stride, mask, (const bf16_t*)feat, (bf16_t*)y, H, vocab, sh, rng);
    return VOX_OK;
}

__global__ __launch_bounds__(256) void k_kv_append(bf16_t* kv, const bf16_t* k, const bf16_t* v, const int* page,
                                                   const int* slot, int page_size, int HD8) {
    const int n = blockIdx.y;
    const int pg = page[n];
    if (pg < 0) return;
    const size_t ps8 = (size_t)2 * page_size * HD8;
    uint4* kd = reinterpret_cast<uint4*>(kv) + (size_t)pg * ps8 + (size_t)slot[n] * HD8;
    uint4* vd = kd + (size_t)page_size * HD8;
    const uint4* ks = reinterpret_cast<const uint4*>(k) + (size_t)n * HD8;
    const uint4* vs = reinterpret_cast<const uint4*>(v) + (size_t)n * HD8;
    for (int i = threadIdx.x; i < HD8; i += 256) {
        kd[i] = ks[i];
        vd[i] = vs[i];
    }
}
int vox_launch_kv_append(hipStream_t st, void* kv, const void* k, const void* v, const int* page, const int* slot,
                         int N, int page_size, int Hkv, int D) {
    if (N <= 0) return VOX_OK;
    if ((Hkv * D) % 8) return vox_fail(VOX_ERR_INVALID, "kv_append: Hkv*D%8!=0");
    hipLaunchKernelGGL(k_kv_append, dim3(1, N), dim3(256), 0, st, (bf16_t*)kv, (const bf16_t*)k, (const bf16_t*)v,
                       page, slot, page_size, Hkv * D / 8);
    return VOX_OK;
}
