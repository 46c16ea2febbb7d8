// On-device sampler: suppress mask, repetition penalty, greedy argmax (first max on ties), top-k / top-p /
// min-p multinomial from a Philox4x32-10 stream, fused with the embedding gather of the sampled id.
// One 256-thread block per logits row.  Replaces vox_serve/sampling.py + the python per-request loops of
// Qwen3TTSModel.sampling / depth_sampling (model/qwen3_tts.py:1863-2004).  Integer/bit-exact contract:
// oracle/voxref.c::vr_argmax / vr_sample.
#include "vox_internal.h"

#define SAMP_KMAX 256
#define SAMP_LDS_VMAX 32768

// (key_of / bits_of / philox_u32: vox_device.h — the persistent depth step's fused sampled pick uses them too)

struct SampArgs {
    bf16_t* logits;
    const int* suppress_ids;
    const uint8_t* rep_cache;
    const uint64_t* offset_dev;
    int* out_ids;
    const bf16_t *emb_table, *emb2_table;
    bf16_t *emb_dst, *feat_acc, *emb2_dst;
    long emb2_dst_stride;
    int H2;
    uint64_t seed, offset, offset_mul;
    long emb_dst_stride;
    float top_p, min_p, temperature, penalty;
    int V, n_suppress, W, C, greedy, top_k, out_stride, out_col, emb_vocab, H, feat_init;
    u32* ws_hist;        // [rows][65536], all-zero between launches (bucket mode)
    uint16_t* ws_keys;   // [rows][SAMP_WS_VMAX] (bucket mode; top-k mode when V > 32768)
};

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u32 lo = __shfl_xor((u32)v, off, VOX_WAVE), hi = __shfl_xor((u32)(v >> 32), off, VOX_WAVE);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

// write the sampled id and (optionally) gather its embedding row / accumulate the next-frame feature
__device__ __forceinline__ void emit_pick(const SampArgs& a, int b, int picked, int tid, int nthr) {
    if (tid == 0) a.out_ids[(size_t)b * a.out_stride + a.out_col] = picked;
    if (a.emb_table) {
        int id = picked < 0 ? 0 : (picked >= a.emb_vocab ? a.emb_vocab - 1 : picked);
        const uint4* src = reinterpret_cast<const uint4*>(a.emb_table + (size_t)id * a.H);
        uint4* dst = a.emb_dst ? reinterpret_cast<uint4*>(a.emb_dst + (size_t)b * a.emb_dst_stride) : nullptr;
        uint4* fa = a.feat_acc ? reinterpret_cast<uint4*>(a.feat_acc + (size_t)b * a.H) : nullptr;
        for (int i = tid; i < (a.H >> 3); i += nthr) {
            const uint4 e = src[i];
            if (dst) dst[i] = e;
            if (fa) {
                uint4 f = a.feat_init ? make_uint4(0, 0, 0, 0) : fa[i];
                uint4 o;
                o.x = (u32)f2bf(bflo(f.x) + bflo(e.x)) | ((u32)f2bf(bfhi(f.x) + bfhi(e.x)) << 16);
                o.y = (u32)f2bf(bflo(f.y) + bflo(e.y)) | ((u32)f2bf(bfhi(f.y) + bfhi(e.y)) << 16);
                o.z = (u32)f2bf(bflo(f.z) + bflo(e.z)) | ((u32)f2bf(bfhi(f.z) + bfhi(e.z)) << 16);
                o.w = (u32)f2bf(bflo(f.w) + bflo(e.w)) | ((u32)f2bf(bfhi(f.w) + bfhi(e.w)) << 16);
                fa[i] = o;
            }
        }
    }
    if (a.emb2_table) {
        const int id = picked < 0 ? 0 : (picked >= a.emb_vocab ? a.emb_vocab - 1 : picked);
        const uint4* src = reinterpret_cast<const uint4*>(a.emb2_table + (size_t)id * a.H2);
        uint4* dst = reinterpret_cast<uint4*>(a.emb2_dst + (size_t)b * a.emb2_dst_stride);
        for (int i = tid; i < (a.H2 >> 3); i += nthr) dst[i] = src[i];
    }
}

// suppress mask + repetition penalty, in place on the row (sampling.py:101-127, qwen3_tts.py:1894-1895)
__device__ __forceinline__ void preprocess_row(const SampArgs& a, bf16_t* lg, int b, int tid, int nthr) {
    if (a.n_suppress > 0) {
        for (int j = tid; j < a.n_suppress; j += nthr) lg[a.suppress_ids[j]] = 0xFF7F;
        __syncthreads();
    }
    if (a.rep_cache && a.penalty != 1.0f) {
        for (int v = tid; v < a.V; v += nthr) {
            int m = 0;
            for (int w = 0; w < a.W; ++w) m |= a.rep_cache[(((size_t)b * a.W + w) * a.C) * a.V + v];
            if (m) {
                const float l = bf2f(lg[v]);
                lg[v] = f2bf(l > 0.0f ? l / a.penalty : l * a.penalty);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_sample(SampArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned long long red[4];
    __shared__ int hist[256];
    __shared__ int sh_i[8];
    __shared__ u32 cand_key[SAMP_KMAX], srt_key[SAMP_KMAX];
    __shared__ int cand_idx[SAMP_KMAX], srt_idx[SAMP_KMAX];
    __shared__ float pe[SAMP_KMAX];
    __shared__ int wcnt[8];     // per-wave tie counts, double-buffered by pass parity

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.V;
    bf16_t* lg = a.logits + (size_t)b * V;

    preprocess_row(a, lg, b, tid, 256);

    int picked;
    if (a.greedy) {
        // (the maximum of (key, ~index) does not depend on the order of the scan: the row is read in 16-byte chunks — a 169k-wide
        // vocabulary, GLM-4-Voice's, took 176 us at one bf16 per thread per pass — with scalar passes up to the first aligned element
        // and past the last whole chunk)
        unsigned long long best = 0;
        auto take = [&](bf16_t x, int v) {
            const unsigned long long c = ((unsigned long long)key_of(x) << 32) | (u32)(0xFFFFFFFFu - (u32)v);
            best = c > best ? c : best;
        };
        int head = (int)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(lg) & 15u)) & 15u) >> 1);
        head = head < V ? head : V;
        for (int v = tid; v < head; v += 256) take(lg[v], v);
        const int nvec = (V - head) >> 3;
        const uint4* lv = reinterpret_cast<const uint4*>(lg + head);
#pragma unroll 4
        for (int i = tid; i < nvec; i += 256) {
            const uint4 q = lv[i];
            const int v0 = head + 8 * i;
            take((bf16_t)(q.x & 0xffff), v0); take((bf16_t)(q.x >> 16), v0 + 1);
            take((bf16_t)(q.y & 0xffff), v0 + 2); take((bf16_t)(q.y >> 16), v0 + 3);
            take((bf16_t)(q.z & 0xffff), v0 + 4); take((bf16_t)(q.z >> 16), v0 + 5);
            take((bf16_t)(q.w & 0xffff), v0 + 6); take((bf16_t)(q.w >> 16), v0 + 7);
        }
        for (int v = head + 8 * nvec + tid; v < V; v += 256) take(lg[v], v);
        best = wave_max_u64(best);
        if (lane == 0) red[wave] = best;
        __syncthreads();
        unsigned long long m = red[0];
        for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
        picked = (int)(0xFFFFFFFFu - (u32)m);
    } else {
        uint16_t* keys = V <= SAMP_LDS_VMAX ? reinterpret_cast<uint16_t*>(smem) : a.ws_keys + (size_t)b * SAMP_WS_VMAX;
        const int k = a.top_k < V ? a.top_k : V;
        hist[tid] = 0;
        __syncthreads();
        for (int v = tid; v < V; v += 256) {
            const u32 ky = key_of(f2bf(bf2f(lg[v]) / a.temperature));
            keys[v] = (uint16_t)ky;
            atomicAdd(&hist[ky >> 8], 1);
        }
        __threadfence_block();   // keys may live in global scratch (V > SAMP_LDS_VMAX)
        __syncthreads();
        if (tid == 0) {
            int cum = 0, b1 = 255;
            for (; b1 > 0; --b1) {
                if (cum + hist[b1] >= k) break;
                cum += hist[b1];
            }
            sh_i[0] = b1;
            sh_i[1] = cum;  // elements strictly above bin b1
        }
        __syncthreads();
        const int b1 = sh_i[0], above1 = sh_i[1];
        __syncthreads();
        hist[tid] = 0;
        __syncthreads();
        for (int v = tid; v < V; v += 256)
            if ((keys[v] >> 8) == b1) atomicAdd(&hist[keys[v] & 0xff], 1);
        __syncthreads();
        if (tid == 0) {
            int cum = above1, b2 = 255;
            for (; b2 > 0; --b2) {
                if (cum + hist[b2] >= k) break;
                cum += hist[b2];
            }
            sh_i[2] = (b1 << 8) | b2;
            sh_i[3] = k - cum;                 // r: ties to take (by ascending index)
            sh_i[4] = hist[b2] > (k - cum);    // more ties than needed -> ordered selection
            sh_i[5] = 0;                       // candidate counter
            sh_i[6] = 0;                       // running tie count
        }
        __syncthreads();
        const u32 T = (u32)sh_i[2];
        const int r = sh_i[3], need_order = sh_i[4];
        if (!need_order) {
            for (int v = tid; v < V; v += 256) {
                const u32 ky = keys[v];
                if (ky >= T) {
                    const int s = atomicAdd(&sh_i[5], 1);
                    cand_key[s] = ky;
                    cand_idx[s] = v;
                }
            }
        } else {
            // the running tie count lives in a register of every thread (all threads add the same four wave counts), and the
            // wave counts are double-buffered by iteration parity: one barrier per pass, and no wave can overwrite a count
            // that a slower wave has yet to read
            int seen = 0;
            for (int base = 0, it = 0; base < V; base += 256, it ^= 1) {
                const int v = base + tid;
                const u32 ky = v < V ? keys[v] : 0;
                const bool tie = v < V && ky == T;
                const unsigned long long bal = __ballot(tie);
                if (lane == 0) wcnt[it * 4 + wave] = __popcll(bal);
                __syncthreads();
                int before = seen;
                for (int w = 0; w < wave; ++w) before += wcnt[it * 4 + w];
                before += __popcll(bal & ((1ull << lane) - 1ull));
                const bool sel = (v < V && ky > T) || (tie && before < r);
                if (sel) {
                    const int s = atomicAdd(&sh_i[5], 1);
                    cand_key[s] = ky;
                    cand_idx[s] = v;
                }
                seen += wcnt[it * 4 + 0] + wcnt[it * 4 + 1] + wcnt[it * 4 + 2] + wcnt[it * 4 + 3];
            }
        }
        __syncthreads();
        const int n0 = sh_i[5];  // == k
        for (int j = tid; j < n0; j += 256) {
            const u32 kj = cand_key[j];
            const int ij = cand_idx[j];
            int rank = 0;
            for (int i = 0; i < n0; ++i) rank += (cand_key[i] > kj) || (cand_key[i] == kj && cand_idx[i] < ij);
            srt_key[rank] = kj;
            srt_idx[rank] = ij;
        }
        __syncthreads();
        const float mval = bf2f(bits_of(srt_key[0]));
        for (int j = tid; j < n0; j += 256) pe[j] = exp2_c((bf2f(bits_of(srt_key[j])) - mval) * VOX_LOG2E);
        __syncthreads();
        if (tid == 0) {
            int n = n0;
            float tot = 0.0f;
            for (int j = 0; j < n; ++j) tot = tot + pe[j];
            if (a.min_p > 0.0f) {
                int kk = 0;
                while (kk < n && pe[kk] >= a.min_p * pe[0]) ++kk;
                n = kk;
                tot = 0.0f;
                for (int j = 0; j < n; ++j) tot = tot + pe[j];
            }
            if (a.top_p < 1.0f) {
                float c = 0.0f;
                const float thr = a.top_p * tot;
                int kk = 0;
                while (kk < n) {
                    c = c + pe[kk];
                    ++kk;
                    if (c >= thr) break;
                }
                n = kk;
                tot = c;
            }
            const uint64_t off = a.offset + (a.offset_dev ? (*a.offset_dev) * a.offset_mul : 0ull);
            const float u = (float)(philox_u32(a.seed, off, (u32)b) >> 8) * (1.0f / 16777216.0f);
            const float thr = u * tot;
            float c = 0.0f;
            int pick = n - 1;
            for (int j = 0; j < n; ++j) {
                c = c + pe[j];
                if (c > thr) {
                    pick = j;
                    break;
                }
            }
            sh_i[7] = srt_idx[pick];
        }
        __syncthreads();
        picked = sh_i[7];
    }
    emit_pick(a, b, picked, tid, 256);
}

// ---- top-k for vocabularies of <= 8192 entries (every codec vocabulary of the hot path): keys in registers ---------------------------
// Same contract as k_sample's top-k branch (oracle/voxref.c::vr_sample: the k largest by (value desc, index asc), exp2 / sums in that order,
// one Philox draw), other mechanics: a thread keeps EPT CONSECUTIVE entries of the row as sortable keys in registers (index order = thread
// order), the k-th largest key is found by bisection on the key value with a block-wide count per probe (16 probes, no atomics — the two
// LDS-atomic histogram passes of k_sample serialise on the few exponent bins a row of logits falls into, and thread 0 then walks 2 x 256
// bins), ties of that key are taken in index order through ONE packed prefix scan (ties | aboves << 16), which also gives every thread the
// slots of its candidates.  The reference's default for Qwen3-TTS is top-k 50 at temperature 0.9 (qwen3_tts.py:1088-1096): 16 sampler calls
// per frame, 23 -> ~8 us each.
template <int EPT>
__global__ __launch_bounds__(256) void k_sample_topk(SampArgs a) {
    __shared__ int wsum[2][4];
    __shared__ int sh_i[8];
    __shared__ u32 cand_key[SAMP_KMAX], srt_key[SAMP_KMAX];
    __shared__ int srt_idx[SAMP_KMAX];
    __shared__ float pe[SAMP_KMAX];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.V;
    bf16_t* lg = a.logits + (size_t)b * V;
    preprocess_row(a, lg, b, tid, 256);
    const int k = a.top_k < V ? a.top_k : V;
    // keys of entries tid * EPT .. + EPT - 1 (0 with valid = false past the row's end)
    u32 key[EPT];
    const int v0 = tid * EPT;
    {
        const bool vec = (reinterpret_cast<uintptr_t>(lg) & 15u) == 0 && v0 + EPT <= V;
        bf16_t raw[EPT];
        if (vec) {
#pragma unroll
            for (int q = 0; q < EPT / 8; ++q) {
                const uint4 w = reinterpret_cast<const uint4*>(lg + v0)[q];
                raw[8 * q + 0] = (bf16_t)(w.x & 0xffff); raw[8 * q + 1] = (bf16_t)(w.x >> 16); raw[8 * q + 2] = (bf16_t)(w.y & 0xffff); raw[8 * q + 3] = (bf16_t)(w.y >> 16);
                raw[8 * q + 4] = (bf16_t)(w.z & 0xffff); raw[8 * q + 5] = (bf16_t)(w.z >> 16); raw[8 * q + 6] = (bf16_t)(w.w & 0xffff); raw[8 * q + 7] = (bf16_t)(w.w >> 16);
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) raw[i] = v0 + i < V ? lg[v0 + i] : (bf16_t)0;
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) key[i] = key_of(f2bf(bf2f(raw[i]) / a.temperature));
    }
    const int nvalid = V - v0 < 0 ? 0 : (V - v0 < EPT ? V - v0 : EPT);
    // block-wide sum of a per-thread count (buffers alternate: one barrier per call)
    // block-wide count of the entries with key >= t: a ballot + popcount per register (scalar arithmetic — as a per-lane count the wave sum is
    // six dependent cross-lane moves per probe, 16 probes), one LDS word per wave, buffers alternating by probe: one barrier per probe
    int par = 0;
    auto count_ge = [&](u32 t) {
        int c = 0;
#pragma unroll
        for (int i = 0; i < EPT; ++i) c += __popcll(__ballot(i < nvalid && key[i] >= t));
        if (lane == 0) wsum[par][wave] = c;
        __syncthreads();
        const int tot = wsum[par][0] + wsum[par][1] + wsum[par][2] + wsum[par][3];
        par ^= 1;
        return tot;
    };
    // the largest T with count(key >= T) >= k  (count(key >= 0) = V >= k)
    u32 lo = 0, hi = 65536;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (count_ge(mid) >= k) lo = mid; else hi = mid;
    }
    const u32 T = lo;
    // ties (key == T) and aboves (key > T) of this thread; one packed inclusive scan over the block in thread order
    int ties = 0, aboves = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        ties += (i < nvalid && key[i] == T) ? 1 : 0;
        aboves += (i < nvalid && key[i] > T) ? 1 : 0;
    }
    int inc = ties | (aboves << 16);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, VOX_WAVE);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[par][wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[par][w];
    const int total = wsum[par][0] + wsum[par][1] + wsum[par][2] + wsum[par][3];
    par ^= 1;
    const int excl = base + inc - (ties | (aboves << 16));
    const int tie_before = excl & 0xffff, above_before = excl >> 16;
    const int above_total = total >> 16;
    const int r = k - above_total;                       // ties to take, by ascending index
    int slot = above_before + (tie_before < r ? tie_before : r);
    int tb = tie_before;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (i < nvalid) {
            const bool isT = key[i] == T;
            if (key[i] > T || (isT && tb < r)) {
                // (key, index) as ONE word whose unsigned order is the candidate order (key descending, index ascending; index < 4096):
                // the rank loop below is then one compare per candidate on unconditional, vectorisable LDS reads — as two arrays with a
                // short-circuit `||` every candidate cost two LDS round trips waited for one by one (5 us of the kernel at k = 50)
                cand_key[slot] = (key[i] << 16) | (0xFFFFu - (u32)(v0 + i));
                ++slot;
            }
            tb += isT ? 1 : 0;
        }
    }
    __syncthreads();
    const int n0 = k;
    for (int j = tid; j < n0; j += 256) {
        const u32 cj = cand_key[j];
        int rank = 0;
        for (int i = 0; i < n0; ++i) rank += cand_key[i] > cj ? 1 : 0;
        srt_key[rank] = cj >> 16;
        srt_idx[rank] = (int)(0xFFFFu - (cj & 0xFFFFu));
    }
    __syncthreads();
    const float mval = bf2f(bits_of(srt_key[0]));
    for (int j = tid; j < n0; j += 256) pe[j] = exp2_c((bf2f(bits_of(srt_key[j])) - mval) * VOX_LOG2E);
    __syncthreads();
    if (wave == 0) {
        const uint64_t off = a.offset + (a.offset_dev ? (*a.offset_dev) * a.offset_mul : 0ull);
        const float u = (float)(philox_u32(a.seed, off, (u32)b) >> 8) * (1.0f / 16777216.0f);
        const int pick = sample_tail_wave(pe, n0, a.min_p, a.top_p, u, lane);
        if (lane == 0) sh_i[7] = srt_idx[pick];
    }
    __syncthreads();
    emit_pick(a, b, sh_i[7], tid, 256);
}

// ---- full-vocabulary ("bucket") mode: top-p-only / min-p-only over any V -------------------------
// Contract: oracle/voxref.c::sample_bucket / bk_find.  One 1024-thread block per row.  The 65536-bin key
// histogram lives in a per-context global scratch (L2-resident, 256 KiB per row) that is all-zero between
// launches; the sortable keys are kept beside it so later passes do not redo the temperature division.
struct BK {
    const u32* hist;
    float m, min_cut;
    int lim_key;
    u32 lim_cnt;
};
__device__ __forceinline__ u32 ld_l2(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float bk_p(const BK& c, u32 key) { return exp2_c((bf2f(bits_of(key)) - c.m) * VOX_LOG2E); }
__device__ __forceinline__ u32 bk_count(const BK& c, u32 key, u32 raw) {
    if ((int)key < c.lim_key) return 0;
    return (int)key == c.lim_key ? c.lim_cnt : raw;
}
__device__ __forceinline__ float bk_mass(const BK& c, u32 key, u32 n) {
    if (!n) return 0.0f;
    const float p = bk_p(c, key);
    if (c.min_cut > 0.0f && p < c.min_cut) return 0.0f;
    return (float)n * p;
}
__device__ __forceinline__ bool bk_cmp(float a, float thr, int strict) { return strict ? a > thr : a >= thr; }

struct BKShared {
    float Cc[256];   // coarse masses of the current (possibly restricted) histogram
    u32 fcnt[256];   // counts of the bin being searched
    float fm[256];   // masses of the bin being searched
    int hf, whole, kp;
    u32 j;
    float acc, tot;
};

// all threads of the block call this; result in S.kp / S.j / S.tot
__device__ void bk_find_blk(const BK& c, float thr, int strict, BKShared& S, int tid) {
    if (tid == 0) {
        float acc = 0.0f;
        int hf = -1, last_h = -1;
        for (int h = 255; h >= 0; --h) {
            const float ch = S.Cc[h];
            if (ch > 0.0f) {
                last_h = h;
                if (bk_cmp(acc + ch, thr, strict)) {
                    hf = h;
                    break;
                }
            }
            acc = acc + ch;
        }
        S.whole = 0;
        if (hf < 0) {
            hf = last_h;
            S.whole = 1;
            acc = 0.0f;
        }
        S.hf = hf;
        S.acc = acc;
    }
    __syncthreads();
    const int hf = S.hf;
    if (tid < 256) {
        const u32 key = (u32)(hf << 8 | tid);
        const u32 n = bk_count(c, key, ld_l2(c.hist + key));
        S.fcnt[tid] = n;
        S.fm[tid] = bk_mass(c, key, n);
    }
    __syncthreads();
    if (tid == 0) {
        float acc2 = S.acc;
        int kp = -1, last_k = -1;
        const int whole = S.whole;
        for (int f = 255; f >= 0; --f) {
            const float mk = S.fm[f];
            if (mk > 0.0f) {
                last_k = hf << 8 | f;
                if (!whole && bk_cmp(acc2 + mk, thr, strict)) {
                    kp = hf << 8 | f;
                    break;
                }
                acc2 = acc2 + mk;
            }
        }
        if (kp < 0) {
            S.kp = last_k;
            S.j = S.fcnt[last_k & 255];
            S.tot = acc2;
        } else {
            const u32 n = S.fcnt[kp & 255];
            u32 lo = 1, hi = n;
            const float p = bk_p(c, (u32)kp);
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (bk_cmp(acc2 + (float)mid * p, thr, strict)) hi = mid;
                else lo = mid + 1;
            }
            S.kp = kp;
            S.j = lo;
            S.tot = acc2 + (float)lo * p;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_sample_bucket(SampArgs a) {
    // pass 1's histogram of one sign's 32768 keys (128 KiB); the coarse pass's staging tile reuses its head afterwards
    __shared__ __attribute__((aligned(16))) u32 lhist[32768];
    u32 (*tile)[33] = reinterpret_cast<u32 (*)[33]>(lhist);
    __shared__ BKShared S;
    __shared__ u32 wred[16];
    __shared__ int sh_pick;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.V;
    bf16_t* lg = a.logits + (size_t)b * V;
    u32* hist = a.ws_hist + (size_t)b * 65536;
    uint16_t* keys = a.ws_keys + (size_t)b * SAMP_WS_VMAX;

    preprocess_row(a, lg, b, tid, 1024);

    // pass 1: keys, histogram, largest key.  Large vocabularies: the histogram is built in LDS, one sign of the values at a time (32768 keys
    // each), and written to the global 65536-bin table with plain stores of the non-zero bins — as 168960 global atomics per row it was 160
    // of the kernel's 300 us (a row of logits falls into a few thousand bins, and same-address atomics serialise in L2); the second half
    // is counted from the keys the first one stored (no second division).  300 -> 207 us per call at 168960 entries.  Small vocabularies
    // keep the global atomics (the two 128 KiB LDS sweeps per half would cost more than they save).
    // (Measured and rejected: the coarse masses summed straight from the LDS histogram — 64-way bank conflicts on 2 of 16 waves: +30 us.)
    u32 kmx = 0;
    if (V >= 32768) {
        for (int half = 1; half >= 0; --half) {
            for (int i = tid; i < 8192; i += 1024) reinterpret_cast<uint4*>(lhist)[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
            if (half) {
                // (the row in 16-byte chunks, two in flight per thread)
                auto take = [&](bf16_t x, int v) {
                    const u32 ky = key_of(f2bf(bf2f(x) / a.temperature));
                    keys[v] = (uint16_t)ky;
                    kmx = ky > kmx ? ky : kmx;
                    if (ky >> 15) atomicAdd(&lhist[ky & 0x7FFFu], 1u);
                };
                auto take8 = [&](const uint4 q, int v0) {
                    take((bf16_t)(q.x & 0xffff), v0); take((bf16_t)(q.x >> 16), v0 + 1); take((bf16_t)(q.y & 0xffff), v0 + 2); take((bf16_t)(q.y >> 16), v0 + 3);
                    take((bf16_t)(q.z & 0xffff), v0 + 4); take((bf16_t)(q.z >> 16), v0 + 5); take((bf16_t)(q.w & 0xffff), v0 + 6); take((bf16_t)(q.w >> 16), v0 + 7);
                };
                int head = (int)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(lg) & 15u)) & 15u) >> 1);
                head = head < V ? head : V;
                for (int v = tid; v < head; v += 1024) take(lg[v], v);
                const int nvec = (V - head) >> 3;
                const uint4* lv = reinterpret_cast<const uint4*>(lg + head);
                int i = tid;
                for (; i + 1024 < nvec; i += 2048) {
                    const uint4 q0 = lv[i], q1 = lv[i + 1024];
                    take8(q0, head + 8 * i);
                    take8(q1, head + 8 * (i + 1024));
                }
                if (i < nvec) take8(lv[i], head + 8 * i);
                for (int v = head + 8 * nvec + tid; v < V; v += 1024) take(lg[v], v);
            } else {
                const int nk = V >> 3;
                const uint4* kv = reinterpret_cast<const uint4*>(keys);
                auto cnt = [&](u32 ky) { if (!(ky >> 15)) atomicAdd(&lhist[ky], 1u); };
                for (int i = tid; i < nk; i += 1024) {
                    const uint4 q = kv[i];
                    cnt(q.x & 0xffff); cnt(q.x >> 16); cnt(q.y & 0xffff); cnt(q.y >> 16); cnt(q.z & 0xffff); cnt(q.z >> 16); cnt(q.w & 0xffff); cnt(q.w >> 16);
                }
                for (int v = 8 * nk + tid; v < V; v += 1024) cnt(keys[v]);
            }
            __syncthreads();
            for (int i = tid; i < 32768; i += 1024) {
                const u32 n = lhist[i];
                if (n) hist[(half << 15) | i] = n;
            }
            __syncthreads();
        }
    } else {
        for (int v = tid; v < V; v += 1024) {
            const u32 ky = key_of(f2bf(bf2f(lg[v]) / a.temperature));
            keys[v] = (uint16_t)ky;
            atomicAdd(hist + ky, 1u);
            kmx = ky > kmx ? ky : kmx;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u32 o = __shfl_xor(kmx, off, VOX_WAVE);
        kmx = o > kmx ? o : kmx;
    }
    if (lane == 0) wred[wave] = kmx;
    __threadfence();
    __syncthreads();
    kmx = wred[0];
    for (int w = 1; w < 16; ++w) kmx = wred[w] > kmx ? wred[w] : kmx;

    BK c;
    c.hist = hist; c.m = bf2f(bits_of(kmx)); c.min_cut = a.min_p > 0.0f ? a.min_p * 1.0f : 0.0f;
    c.lim_key = -1; c.lim_cnt = 0;

    // pass 2: coarse masses, thread h sums its 256 keys in descending order; tiles staged coalesced
    float csum = 0.0f;
    for (int t = 7; t >= 0; --t) {
        __syncthreads();
        for (int e = tid; e < 8192; e += 1024) {
            const int h = e >> 5, f = e & 31;
            tile[h][f] = ld_l2(hist + (h << 8) + (t << 5) + f);
        }
        __syncthreads();
        if (tid < 256)
            for (int f = 31; f >= 0; --f) csum = csum + bk_mass(c, (u32)(tid << 8 | t << 5 | f), tile[tid][f]);
    }
    if (tid < 256) S.Cc[tid] = csum;
    __syncthreads();

    float tot;
    if (tid == 0) {
        float t0 = 0.0f;
        for (int h = 255; h >= 0; --h) t0 = t0 + S.Cc[h];
        S.tot = t0;
    }
    __syncthreads();
    tot = S.tot;
    if (a.top_p < 1.0f) {
        bk_find_blk(c, a.top_p * tot, 0, S, tid);
        c.lim_key = S.kp;
        c.lim_cnt = S.j;
        tot = S.tot;
        __syncthreads();
        // restricted coarse masses: bins below the limit vanish, the limit's bin is re-summed
        const int lh = c.lim_key >> 8;   // == S.hf: fcnt still holds that bin's raw counts
        if (tid < 256) {
            const u32 key = (u32)(lh << 8 | tid);
            S.fm[tid] = bk_mass(c, key, bk_count(c, key, S.fcnt[tid]));
            if (tid < lh) S.Cc[tid] = 0.0f;
        }
        __syncthreads();
        if (tid == 0) {
            float s2 = 0.0f;
            for (int f = 255; f >= 0; --f) s2 = s2 + S.fm[f];
            S.Cc[lh] = s2;
        }
        __syncthreads();
    }
    const uint64_t off = a.offset + (a.offset_dev ? (*a.offset_dev) * a.offset_mul : 0ull);
    const float u = (float)(philox_u32(a.seed, off, (u32)b) >> 8) * (1.0f / 16777216.0f);
    bk_find_blk(c, u * tot, 1, S, tid);
    const u32 kp = (u32)S.kp, j = S.j;
#if defined(BK_STOP) && BK_STOP == 3
    if (kmx != 0x12345u) return;
#endif

    // pass 3: the j-th element (ascending index) whose key is kp; wave w owns a contiguous index range
    const int chunk = ((V + 15) / 16 + 63) & ~63;
    const int lo = wave * chunk, hi = min(V, lo + chunk);
    u32 cnt = 0;
    for (int base = lo; base < hi; base += 64) {
        const int v = base + lane;
        cnt += __popcll(__ballot(v < hi && keys[v] == kp));
    }
    if (lane == 0) wred[wave] = cnt;
    if (tid == 0) sh_pick = -1;
    __syncthreads();
    u32 before = 0;
    for (int w = 0; w < wave; ++w) before += wred[w];
    if (before < j && j <= before + cnt) {
        u32 run = before;
        for (int base = lo; base < hi; base += 64) {
            const int v = base + lane;
            const bool mt = v < hi && keys[v] == kp;
            const unsigned long long bal = __ballot(mt);
            const u32 mine = run + __popcll(bal & ((1ull << lane) - 1ull)) + 1;
            if (mt && mine == j) sh_pick = v;
            run += __popcll(bal);
            if (run >= j) break;
        }
    }
    // leave the histogram all-zero for the next launch
    uint4* hz = reinterpret_cast<uint4*>(hist);
    for (int i = tid; i < 16384; i += 1024) hz[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    emit_pick(a, b, sh_pick, tid, 1024);
}

int vox_launch_sample(hipStream_t st, const SampleCall& c) {
    if (c.B <= 0) return VOX_OK;
    // top_k <= 0 with top_p >= 1 and min_p <= 0 is the full multinomial (top_p_sampling(p=1.0), sampling.py:41-52)
    const bool greedy = c.cfg.greedy || c.cfg.temperature == 0.0f;
    SampArgs a{};
    a.logits = (bf16_t*)c.logits; a.suppress_ids = c.suppress_ids; a.rep_cache = c.rep_cache;
    a.offset_dev = c.offset_dev; a.out_ids = c.out_ids; a.emb_table = (const bf16_t*)c.emb_table;
    a.emb_dst = (bf16_t*)c.emb_dst; a.feat_acc = (bf16_t*)c.feat_acc; a.seed = c.seed; a.offset = c.offset;
    a.offset_mul = c.offset_mul; a.emb_dst_stride = c.emb_dst_stride; a.top_p = c.cfg.top_p;
    a.min_p = c.cfg.min_p; a.temperature = c.cfg.temperature; a.penalty = c.cfg.repetition_penalty;
    a.V = c.V; a.n_suppress = c.n_suppress; a.W = c.W; a.C = c.C; a.greedy = greedy ? 1 : 0;
    a.top_k = c.cfg.top_k; a.out_stride = c.out_stride; a.out_col = c.out_col; a.emb_vocab = c.emb_vocab;
    a.H = c.H; a.feat_init = c.feat_init;
    a.emb2_table = (const bf16_t*)c.emb2_table; a.emb2_dst = (bf16_t*)c.emb2_dst; a.emb2_dst_stride = c.emb2_dst_stride; a.H2 = c.H2;
    if (c.emb2_table && ((c.H2 % 8) || !c.emb2_dst || c.emb_vocab <= 0)) return vox_fail(VOX_ERR_INVALID, "sample: bad second gather");
    a.ws_hist = (u32*)c.ws;
    a.ws_keys = c.ws ? reinterpret_cast<uint16_t*>((char*)c.ws + (size_t)SAMP_WS_ROWS * 65536 * 4) : nullptr;
    if (c.emb_table && (c.H % 8)) return vox_fail(VOX_ERR_INVALID, "sample: H%8!=0");
    size_t smem = 0;
    if (!greedy) {
        if (c.cfg.min_p > 1.0f) return vox_fail(VOX_ERR_INVALID, "sample: min_p > 1");
        const bool bucket = c.cfg.top_k <= 0;
        if (bucket || c.V > SAMP_LDS_VMAX) {
            if (!c.ws) return vox_fail(VOX_ERR_INVALID, "sample: this mode needs the context's sampler scratch");
            if (c.B > SAMP_WS_ROWS || c.V > SAMP_WS_VMAX)
                return vox_fail(VOX_ERR_INVALID, "sample: B %d > %d or V %d > %d", c.B, SAMP_WS_ROWS, c.V, SAMP_WS_VMAX);
        }
        if (bucket) {
            hipLaunchKernelGGL(k_sample_bucket, dim3(c.B), dim3(1024), 0, st, a);
            return VOX_OK;
        }
        if (c.cfg.top_k > SAMP_KMAX)
            return vox_fail(VOX_ERR_INVALID, "sample: top_k %d > %d", c.cfg.top_k, SAMP_KMAX);
        smem = c.V <= SAMP_LDS_VMAX ? (size_t)c.V * 2 : 0;
        // <= 8192 entries: the register form (VOX_SAMPLE_TOPK_REG=0: the histogram form, A/B)
        static const bool reg = [] { const char* e = getenv("VOX_SAMPLE_TOPK_REG"); return !(e && e[0] == '0'); }();
        if (reg && c.V <= 8192) {      // (Qwen3-TTS 3072 / 2048, CSM-1B 2051, CosyVoice2 6564)
            if (c.V <= 2048) hipLaunchKernelGGL(k_sample_topk<8>, dim3(c.B), dim3(256), 0, st, a);
            else if (c.V <= 4096) hipLaunchKernelGGL(k_sample_topk<16>, dim3(c.B), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(k_sample_topk<32>, dim3(c.B), dim3(256), 0, st, a);
            return VOX_OK;
        }
    }
    hipLaunchKernelGGL(k_sample, dim3(c.B), dim3(256), smem, st, a);
    return VOX_OK;
}

// ---- standalone pieces for the drop-in Sampler API -------------------------------------------------
__global__ void k_suppress(bf16_t* lg, int V, const int* ids, int n) {
    const int b = blockIdx.y;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) lg[(size_t)b * V + ids[j]] = 0xFF7F;
}
int vox_launch_suppress(hipStream_t st, void* logits, int B, int V, const int* ids, int n) {
    if (B <= 0 || n <= 0) return VOX_OK;
    hipLaunchKernelGGL(k_suppress, dim3((n + 255) / 256, B), dim3(256), 0, st, (bf16_t*)logits, V, ids, n);
    return VOX_OK;
}

__global__ void k_rep_penalty(bf16_t* lg, const uint8_t* cache, int W, int C, int V, float p) {
    const int b = blockIdx.y;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
        int m = 0;
        for (int w = 0; w < W; ++w) m |= cache[(((size_t)b * W + w) * C) * V + v];
        if (m) {
            const float l = bf2f(lg[(size_t)b * V + v]);
            lg[(size_t)b * V + v] = f2bf(l > 0.0f ? l / p : l * p);
        }
    }
}
int vox_launch_rep_penalty(hipStream_t st, void* logits, const uint8_t* cache, int B, int W, int C, int V, float p) {
    if (B <= 0) return VOX_OK;
    hipLaunchKernelGGL(k_rep_penalty, dim3((V + 255) / 256, B), dim3(256), 0, st, (bf16_t*)logits, cache, W, C, V, p);
    return VOX_OK;
}

// multi-codebook form: logits [B, Cl, V]; row (b, c) is penalised by cache codebook c
__global__ void k_rep_penalty_mc(bf16_t* lg, const uint8_t* cache, int Cl, int W, int C, int V, float p) {
    const int b = blockIdx.y / Cl, c = blockIdx.y % Cl;
    bf16_t* row = lg + ((size_t)b * Cl + c) * V;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
        int m = 0;
        for (int w = 0; w < W; ++w) m |= cache[(((size_t)b * W + w) * C + c) * V + v];
        if (m) {
            const float l = bf2f(row[v]);
            row[v] = f2bf(l > 0.0f ? l / p : l * p);
        }
    }
}
int vox_launch_rep_penalty_mc(hipStream_t st, void* logits, const uint8_t* cache, int B, int Cl, int W, int C, int V, float p) {
    if (B <= 0 || Cl <= 0) return VOX_OK;
    if (Cl > C) return vox_fail(VOX_ERR_INVALID, "rep_penalty_mc: more logit codebooks than cache codebooks");
    hipLaunchKernelGGL(k_rep_penalty_mc, dim3((V + 255) / 256, B * Cl), dim3(256), 0, st, (bf16_t*)logits, cache, Cl, W, C, V, p);
    return VOX_OK;
}
// every (row b, slot w, plane c) receives every id of the step: ids [B * Cl]
__global__ void k_rep_set_mc(uint8_t* cache, const int* ids, int n_ids, int B, int W, int C, int V, int only_last) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * W * C * n_ids;
    if (i >= total) return;
    const int j = (int)(i % n_ids), c = (int)((i / n_ids) % C), w = (int)((i / ((size_t)n_ids * C)) % W), b = (int)(i / ((size_t)n_ids * C * W));
    if (only_last && w != W - 1) return;
    cache[(((size_t)b * W + w) * C + c) * V + ids[j]] = 1;
}
int vox_launch_rep_update_mc(hipStream_t st, uint8_t* cache, const int* ids, int B, int Cl, int W, int C, int V, int window);

// window shift (one block per batch row), then the leak-faithful set: every row gets every request's id
__global__ void k_rep_shift(uint8_t* cache, int W, int C, int V) {
    uint8_t* cb = cache + (size_t)blockIdx.x * W * C * V;
    const size_t plane = (size_t)C * V;
    for (int w = 0; w + 1 < W; ++w) {
        for (size_t i = threadIdx.x; i < plane; i += blockDim.x) cb[w * plane + i] = cb[(w + 1) * plane + i];
        __syncthreads();
    }
    for (size_t i = threadIdx.x; i < plane; i += blockDim.x) cb[(size_t)(W - 1) * plane + i] = 0;
}
__global__ void k_rep_set(uint8_t* cache, const int* ids, int B, int W, int C, int V, int only_last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over B(rows) * W * B(ids)
    const int total = B * W * B;
    if (i >= total) return;
    const int b = i / (W * B), w = (i / B) % W, j = i % B;
    if (only_last && w != W - 1) return;
    cache[(((size_t)b * W + w) * C) * V + ids[j]] = 1;
}
int vox_launch_rep_update(hipStream_t st, uint8_t* cache, const int* ids, int B, int W, int C, int V, int window) {
    if (B <= 0) return VOX_OK;
    if (window > 1) hipLaunchKernelGGL(k_rep_shift, dim3(B), dim3(256), 0, st, cache, W, C, V);
    const int total = B * W * B;
    hipLaunchKernelGGL(k_rep_set, dim3((total + 255) / 256), dim3(256), 0, st, cache, ids, B, W, C, V,
                       window > 1 ? 1 : 0);
    return VOX_OK;
}
int vox_launch_rep_update_mc(hipStream_t st, uint8_t* cache, const int* ids, int B, int Cl, int W, int C, int V, int window) {
    if (B <= 0 || Cl <= 0) return VOX_OK;
    if (window > 1) hipLaunchKernelGGL(k_rep_shift, dim3(B), dim3(256), 0, st, cache, W, C, V);
    const size_t total = (size_t)B * W * C * B * Cl;
    hipLaunchKernelGGL(k_rep_set_mc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cache, ids, B * Cl, B, W, C, V,
                       window > 1 ? 1 : 0);
    return VOX_OK;
}
