// Shared device helpers for the ShadowKV gfx950 kernels.
//
// The arithmetic in this header is the *contract* that oracle/shadowkv_oracle.c restates
// on the CPU (same operation order, same roundings), so selection results are bit-exact.
// Built with -ffp-contract=off: every fused multiply-add below is written explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SKV_WAVE 64

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- bf16 <-> f32 ---------------------------------------------------------------------
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even, NaN stays NaN (same as oracle f2bf / v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// bf16 arithmetic, one rounding per operation (CUDA __hmul/__hadd semantics,
// /root/reference/kernels/rope_new.cu:366-367)
__device__ __forceinline__ float bfr(float x) { return bf2f(f2bf(x)); }  // round f32 to bf16 grid

// ---- exp contract (oracle spec_exp) ---------------------------------------------------
__device__ __forceinline__ float spec_exp(float x) {
    if (!(x >= -80.0f)) return 0.0f;
    const float LOG2E = 1.44269504088896341f;
    const float LN2_HI = 0.693145751953125f;
    const float LN2_LO = 1.42860682030941723e-6f;
    float t = x * LOG2E;
    float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -LN2_HI, x);
    r = __builtin_fmaf(n, -LN2_LO, r);
    float p = 1.0f / 720.0f;
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    int bits = __float_as_int(p) + (((int)n) << 23);
    return __int_as_float(bits);
}

// e in [0,2) -> trunc(e * 2^36)  (oracle exp_to_fixed)
__device__ __forceinline__ unsigned long long exp_to_fixed(float e) {
    uint32_t bits = __float_as_uint(e);
    int ex = (int)((bits >> 23) & 0xff);
    if (ex == 0) return 0ull;
    unsigned long long mant = (unsigned long long)((bits & 0x7fffffu) | 0x800000u);
    int sh = ex - 127 - 23 + 36;
    if (sh >= 0) return mant << sh;
    if (sh <= -24) return 0ull;
    return mant >> (-sh);
}
__device__ __forceinline__ float fixed_to_float(unsigned long long S) {
    return (float)((double)S * (1.0 / 68719476736.0));
}

// ---- cross-lane helpers (wave64) ------------------------------------------------------
// sum over the 16 lanes of a DPP row as the balanced tree ((p0+p1)+(p2+p3))+... ; every
// lane of the row ends with the same value (IEEE add is commutative).
__device__ __forceinline__ float row16_tree_sum(float v) {
    v = v + __shfl_xor(v, 1, 64);
    v = v + __shfl_xor(v, 2, 64);
    v = v + __shfl_xor(v, 4, 64);
    v = v + __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ float wave_tree_sum(float v) {
    v = v + __shfl_xor(v, 1, 64);
    v = v + __shfl_xor(v, 2, 64);
    v = v + __shfl_xor(v, 4, 64);
    v = v + __shfl_xor(v, 8, 64);
    v = v + __shfl_xor(v, 16, 64);
    v = v + __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o, 64);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// error codes of the C ABI
#define SKV_OK 0
#define SKV_ERR_ARG (-1)
#define SKV_ERR_UNSUPPORTED (-2)
#define SKV_ERR_LAUNCH (-3)
