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

// round-to-nearest-even, NaN stays NaN: one v_cvt_pk_bf16_f32 (same results as the oracle's f2bf
// for every non-NaN input)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// bf16 arithmetic, one rounding per operation (CUDA __hmul/__hadd semantics,
// /root/reference/kernels/rope_new.cu:366-367)
__device__ __forceinline__ float bfr(float x) { return bf2f(f2bf(x)); }  // round f32 to bf16 grid
// order-preserving unsigned 16-bit key of a bf16 (x >= 0: x | 0x8000; x < 0: ~x) - the sampler's keys (skv_sample.hip)
__device__ __forceinline__ uint32_t skv_bf16_order_key(bf16_t b) {
    const uint32_t w = (uint32_t)b;
    return (w ^ ((w & 0x8000u) ? 0xffffu : 0x8000u)) & 0xffffu;
}

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
// DPP row operations (no LDS traffic, unlike __shfl_xor which lowers to ds_bpermute_b32).
// 0xB1 = quad_perm[1,0,3,2] (lane^1), 0x4E = quad_perm[2,3,0,1] (lane^2),
// 0x141 = row_half_mirror (lane i <-> 7-i inside 8 lanes), 0x140 = row_mirror (i <-> 15-i).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row as the balanced tree ((p0+p1)+(p2+p3))+... ; every
// lane of the row ends with the same value (IEEE add is commutative, and after the first two
// steps all lanes of a quad hold the quad sum, so the mirrors deliver "the other half's" sum).
__device__ __forceinline__ float row16_tree_sum(float v) {
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    v = v + dpp_mov<0x141>(v);
    v = v + dpp_mov<0x140>(v);
    return v;
}
// lane i <-> lane i^4 inside a row: two bank-masked row shifts (banks = quads of the row)
__device__ __forceinline__ float dpp_xor4(float v) {
    int x = __float_as_int(v);
    int r = __builtin_amdgcn_update_dpp(0, x, 0x104 /*row_shl:4*/, 0xF, 0x5, false);   // quads 0,2 <- lane+4
    r = __builtin_amdgcn_update_dpp(r, x, 0x114 /*row_shr:4*/, 0xF, 0xA, false);       // quads 1,3 <- lane-4
    return __int_as_float(r);
}
// Reduce NV values per lane over the 16 lanes of a row with the SAME tree as row16_tree_sum
// (xor 1, 2, 4, 8), transposing as it goes: at each of the first log2(NV) stages a lane keeps half of
// its values and hands the other half to its partner, so it finishes with ONE value, the row total
// for index g = row16_owner<NV>(lane).  x[l] + x[l^m] is computed by both partners (commutative), so
// every lane that ends up with the same g holds the same bits.
// per-lane select by a constant 64-bit lane mask: one v_cndmask_b32, no compare (hipcc otherwise
// re-materialises a v_cmp on the lane id for every select)
__device__ __forceinline__ float sel_lanes(float if_clear, float if_set, unsigned long long lane_mask) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(lane_mask));
    return r;
}
template <int NV>
__device__ __forceinline__ float row16_tree_sum_transposed(float (&v)[NV], int lane) {
    (void)lane;
    int nv = NV;
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
        // lanes whose index bit `stage` is set
        const unsigned long long bitmask = stage == 0   ? 0xAAAAAAAAAAAAAAAAull
                                           : stage == 1 ? 0xCCCCCCCCCCCCCCCCull
                                           : stage == 2 ? 0xF0F0F0F0F0F0F0F0ull
                                                        : 0xFF00FF00FF00FF00ull;
        if (nv > 1) {
            const int half = nv / 2;
#pragma unroll
            for (int k = 0; k < NV / 2; ++k) {
                if (k < half) {
                    const float keep = sel_lanes(v[k], v[k + half], bitmask);
                    const float give = sel_lanes(v[k + half], v[k], bitmask);
                    const float got = stage == 0 ? dpp_mov<0xB1>(give)
                                      : stage == 1 ? dpp_mov<0x4E>(give)
                                      : stage == 2 ? dpp_xor4(give) : dpp_mov<0x128>(give);
                    v[k] = keep + got;
                }
            }
            nv = half;
        } else {
            const float got = stage == 0 ? dpp_mov<0xB1>(v[0])
                              : stage == 1 ? dpp_mov<0x4E>(v[0])
                              : stage == 2 ? dpp_xor4(v[0]) : dpp_mov<0x128>(v[0]);
            v[0] = v[0] + got;
        }
    }
    return v[0];
}
// which of the NV values lane `lane` ends up owning, and whether it is the lane that should publish it
template <int NV>
__device__ __forceinline__ int row16_owner(int lane) {
    int g = 0, w = NV / 2;
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
        if (w >= 1) g += ((lane >> stage) & 1) * w;
        w /= 2;
    }
    return g;
}
template <int NV>
__device__ __forceinline__ bool row16_publisher(int lane) {
    // lanes whose non-transposed index bits are zero (one per g and row)
    constexpr int stages = NV >= 16 ? 4 : NV >= 8 ? 3 : NV >= 4 ? 2 : NV >= 2 ? 1 : 0;
    return ((lane & 15) >> stages) == 0;
}
__device__ __forceinline__ float wave_tree_sum(float v) {
    v = row16_tree_sum(v);
    v = v + __shfl_xor(v, 16, 64);
    v = v + __shfl_xor(v, 32, 64);
    return v;
}
// wave-wide max / u64 sum: 4 DPP steps inside each 16-lane row, then the 4 row results through readlane
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    const int x = __float_as_int(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(x, 0)), b = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(x, 32)), d = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_mov_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_sum_u64_dpp(unsigned long long v) {
    v += dpp_mov_u64<0xB1>(v);
    v += dpp_mov_u64<0x4E>(v);
    v += dpp_mov_u64<0x141>(v);
    v += dpp_mov_u64<0x140>(v);
    unsigned long long t = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 16 * r);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * r);
        t += ((unsigned long long)hi << 32) | lo;
    }
    return t;
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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: launchers that need more than 64 KB of
// dynamic LDS remember what they have set per device (a process may drive several GPUs), not in one process-wide flag.
// `set_bytes`: a zero-initialised static array of the call site.  Returns 0 or SKV_ERR_LAUNCH (-3).
static inline int skv_ensure_max_lds(const void* fn, size_t bytes, size_t (&set_bytes)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (bytes <= 64 * 1024 || bytes <= set_bytes[dev]) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -3;
    set_bytes[dev] = bytes;
    return 0;
}

// error codes of the C ABI
#define SKV_OK 0
#define SKV_ERR_ARG (-1)
#define SKV_ERR_UNSUPPORTED (-2)
#define SKV_ERR_LAUNCH (-3)
