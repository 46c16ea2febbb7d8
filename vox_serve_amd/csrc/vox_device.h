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
template <int WIDTH>
__device__ __forceinline__ float butterfly(float s) {
#pragma unroll
    for (int off = WIDTH / 2; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, VOX_WAVE);
    return s;
}

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
// 16-byte non-temporal (streaming) load: weights are read once per step by exactly one wave
__device__ __forceinline__ uint4 ldg_nt(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
