// Device-side helpers shared by the gfx950 kernels: bf16 <-> f32, the fixed-order wave64 reductions and
// the polynomial exp2 of the numeric contract.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef uint32_t u32;

#define VOX_WAVE 64
#define VOX_TC 32
#define VOX_LOG2E 1.44269504088896340736f

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((u32)h) << 16); }
__device__ __forceinline__ float bflo(u32 w) { return __uint_as_float(w << 16); }          // element 0 of a pair
__device__ __forceinline__ float bfhi(u32 w) { return __uint_as_float(w & 0xffff0000u); }  // element 1 of a pair
// fp32 -> bf16, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950 (same results as the integer recipe of
// oracle/voxref.c::f2bf for every non-NaN input; at one wave per SIMD instruction count is time)
typedef __bf16 vox_bf2 __attribute__((ext_vector_type(2)));
typedef float vox_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) {
    union { __bf16 b; bf16_t u; } c;
    c.b = (__bf16)f;
    return c.u;
}
// two values at once: element 0 in the low half
__device__ __forceinline__ u32 pack_bf2(float lo, float hi) {
    const vox_f2 v = {lo, hi};
    union { vox_bf2 b; u32 u; } c;
    c.b = __builtin_convertvector(v, vox_bf2);
    return c.u;
}
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }

// 8 sequential fmaf over one 16-byte chunk pair (elements in memory order)
// A pointer the kernel read from a device-side table (not a kernel argument) has no known address space: every load through it is a
// FLAT load — counted by vmcnt AND lgkmcnt, so the next LDS wait (every barrier has one) also waits for all weight rows in flight.
// Round trip through the global address space: the loads become global_load again.
template <typename T>
__device__ __forceinline__ const T* assume_global(const T* p) {
    return (const T*)(const __attribute__((address_space(1))) T*)p;
}
__device__ __forceinline__ float dot8(uint4 w, uint4 x, float a) {
    a = __fmaf_rn(bflo(w.x), bflo(x.x), a);
    a = __fmaf_rn(bfhi(w.x), bfhi(x.x), a);
    a = __fmaf_rn(bflo(w.y), bflo(x.y), a);
    a = __fmaf_rn(bfhi(w.y), bfhi(x.y), a);
    a = __fmaf_rn(bflo(w.z), bflo(x.z), a);
    a = __fmaf_rn(bfhi(w.z), bfhi(x.z), a);
    a = __fmaf_rn(bflo(w.w), bflo(x.w), a);
    a = __fmaf_rn(bfhi(w.w), bfhi(x.w), a);
    return a;
}
__device__ __forceinline__ float sq8(uint4 x, float a) { return dot8(x, x, a); }

// xor-butterfly over `width` lanes (width power of two <= 64): s = s + s_partner, offsets width/2 .. 1.
// Every participating lane ends with the same value.
#ifndef VOX_DPP_BUTTERFLY
#define VOX_DPP_BUTTERFLY 1
#endif
#if VOX_DPP_BUTTERFLY
// The same partner pairs without the LDS crossbar (__shfl_xor lowers to ds_bpermute_b32, ~2 LDS latencies per level, six levels
// per DOT): lane ^ 32 / ^ 16 via gfx950's v_permlane32_swap / v_permlane16_swap (one swap hands every lane its own and its
// partner's value in the two results; the add is commutative, so no select), lane ^ 8 / ^ 2 / ^ 1 via one DPP move, lane ^ 4 via
// two bank-masked DPP moves.  Bit-identical sums (tools/dpp_check.hip).
typedef u32 vox_u2 __attribute__((ext_vector_type(2)));
template <int CTRL, int BANK>
__device__ __forceinline__ float vox_dpp(float old, float s) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(s), CTRL, 0xF, BANK, false));
}
template <int OFF>
__device__ __forceinline__ float xor_add(float s) {
    if constexpr (OFF == 32) {
        const vox_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        return __uint_as_float(r.x) + __uint_as_float(r.y);
    } else if constexpr (OFF == 16) {
        const vox_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        return __uint_as_float(r.x) + __uint_as_float(r.y);
    } else if constexpr (OFF == 8) {
        return s + vox_dpp<0x128, 0xF>(s, s);                  // row_ror:8
    } else if constexpr (OFF == 4) {
        float t = vox_dpp<0x104, 0x5>(s, s);                   // banks 0, 2 <- lane + 4
        t = vox_dpp<0x114, 0xA>(t, s);                         // banks 1, 3 <- lane - 4
        return s + t;
    } else if constexpr (OFF == 2) {
        return s + vox_dpp<0x4E, 0xF>(s, s);                   // quad_perm [2,3,0,1]
    } else {
        return s + vox_dpp<0xB1, 0xF>(s, s);                   // quad_perm [1,0,3,2]
    }
}
template <int WIDTH>
__device__ __forceinline__ float butterfly(float s) {
    if constexpr (WIDTH >= 64) s = xor_add<32>(s);
    if constexpr (WIDTH >= 32) s = xor_add<16>(s);
    if constexpr (WIDTH >= 16) s = xor_add<8>(s);
    if constexpr (WIDTH >= 8) s = xor_add<4>(s);
    if constexpr (WIDTH >= 4) s = xor_add<2>(s);
    if constexpr (WIDTH >= 2) s = xor_add<1>(s);
    return s;
}
#else
template <int WIDTH>
__device__ __forceinline__ float butterfly(float s) {
#pragma unroll
    for (int off = WIDTH / 2; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, VOX_WAVE);
    return s;
}
#endif

__device__ __forceinline__ float exp2_c(float x) {
    if (!(x > -125.0f)) return 0.0f;
    if (x >= 128.0f) return __uint_as_float(0x7f800000u);
    float n = __builtin_rintf(x);
    float f = x - n;
    float p = 1.52527338e-5f;
    p = __fmaf_rn(p, f, 1.54035304e-4f);
    p = __fmaf_rn(p, f, 1.33335581e-3f);
    p = __fmaf_rn(p, f, 9.61812911e-3f);
    p = __fmaf_rn(p, f, 5.55041087e-2f);
    p = __fmaf_rn(p, f, 2.40226507e-1f);
    p = __fmaf_rn(p, f, 6.93147181e-1f);
    p = __fmaf_rn(p, f, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (((u32)(int)n) << 23));
}
__device__ __forceinline__ float silu_c(float g) {
    float e = exp2_c((-g) * VOX_LOG2E);
    return g / (1.0f + e);
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 as_uint4(u32x4_t v) { return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 as_uint4(uint4 v) { return v; }
// 16-byte non-temporal (streaming) load: weights are read once per step by exactly one wave
// (through the global address space explicitly: a pointer read from a device-side table has no known address space, and a load
// through it is a FLAT load — counted by vmcnt AND lgkmcnt, so the next LDS wait also waits for every weight row in flight)
#define VOX_GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ uint4 ldg_nt(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load((const VOX_GLOBAL_AS u32x4_t*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// plain 16-byte load of global memory (same reason)
__device__ __forceinline__ uint4 ldg(const uint4* p) {
    const u32x4_t v = *(const VOX_GLOBAL_AS u32x4_t*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- sampler pieces shared by sampler.hip and the persistent depth step's fused pick (kernels_lm.hip) ----
// sortable key of a bf16 value (value order == unsigned key order; -0 == +0) and its inverse; Philox4x32-10 word 0 of (seed, offset, row)
__device__ __forceinline__ u32 key_of(bf16_t b) {
    if (b == 0x8000) b = 0;  // -0 == +0
    return (b & 0x8000) ? (u32)(~b & 0xffff) : (u32)(b | 0x8000);
}
__device__ __forceinline__ bf16_t bits_of(u32 key) { return (key & 0x8000) ? (bf16_t)(key & 0x7fff) : (bf16_t)(~key & 0xffff); }

__device__ __forceinline__ u32 philox_u32(uint64_t seed, uint64_t offset, u32 row) {
    u32 c0 = (u32)offset, c1 = (u32)(offset >> 32), c2 = row, c3 = 0;
    u32 k0 = (u32)seed, k1 = (u32)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0, n1 = (u32)p1;
        const u32 n2 = (u32)(p0 >> 32) ^ c3 ^ k1, n3 = (u32)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

// The serial tail of the top-k samplers (oracle/voxref.c::vr_sample: tot = sum of pe[0..n) IN ORDER, the optional min-p / top-p cuts, the
// first j with pe[0] + .. + pe[j] > u * tot), run by ONE full wave; returns the picked candidate position.  pe: the n0 <= 256 probabilities
// in candidate order (LDS).  The sums have to be taken sequentially (the contract is their rounding), but every decision is a comparison
// against a PREFIX of that one chain: so the chain runs once — candidate j's value comes out of lane j % 64 through v_readlane with a
// compile-time lane, every lane adds (uniform c), lane j keeps c_j — and the cuts and the pick are ballots over the kept prefixes.  (As a
// thread-0 loop over LDS each of the 2 k dependent adds waited for its own ds_read: ~3.4 of the kernel's ~10 us at k = 50.)
__device__ __forceinline__ int sample_tail_wave(const float* pe, int n0, float min_p, float top_p, float u, int lane) {
    float per[4], cpre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) per[q] = (lane + 64 * q) < n0 ? pe[lane + 64 * q] : 0.0f;      // (+ 0.0f leaves a non-negative sum unchanged)
    float c = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float cm = 0.0f;
        if (64 * q < n0) {                  // uniform
#pragma unroll
            for (int J = 0; J < 64; ++J) {
                c = c + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, per[q]), J));
                cm = lane == J ? c : cm;
            }
        }
        cpre[q] = cm;                       // c_j of candidate j = lane + 64 q
    }
    // first candidate position < lim for which flag holds, else lim
    auto first_of = [&](const bool (&f)[4], int lim) {
        int r = lim;
#pragma unroll
        for (int q = 3; q >= 0; --q) {
            const unsigned long long b = __ballot(f[q] && (lane + 64 * q) < lim);
            if (b) r = 64 * q + (int)__builtin_ctzll(b);
        }
        return r;
    };
    auto prefix_at = [&](int j) {           // c_j for a uniform j in [0, n0)
        const int q = j >> 6, l = j & 63;
        float v = 0.0f;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cpre[qq]), __builtin_amdgcn_readfirstlane(l)));
            v = qq == q ? t : v;
        }
        return v;
    };
    int n = n0;
    float tot = prefix_at(n - 1);
    if (min_p > 0.0f) {
        const float cut = min_p * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, per[0]), 0));
        const bool f[4] = {!(per[0] >= cut), !(per[1] >= cut), !(per[2] >= cut), !(per[3] >= cut)};
        n = first_of(f, n);
        tot = n > 0 ? prefix_at(n - 1) : 0.0f;
    }
    if (top_p < 1.0f) {
        const float thr = top_p * tot;
        const bool f[4] = {cpre[0] >= thr, cpre[1] >= thr, cpre[2] >= thr, cpre[3] >= thr};
        const int j = first_of(f, n);
        if (j < n) { n = j + 1; tot = prefix_at(j); }
    }
    const float thr = u * tot;
    const bool f[4] = {cpre[0] > thr, cpre[1] > thr, cpre[2] > thr, cpre[3] > thr};
    const int j = first_of(f, n);
    return j < n ? j : n - 1;
}

