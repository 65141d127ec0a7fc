// bf16 GEMV for the decode step's dense projections (q_len == 1, one token per sequence):
//     y[n] = sum_k W[n][k] * x[k] (+ bias[n]),   W [N][K] row-major (K contiguous), f32 accumulation, bf16 out
// and the gate/up variant with the activation fused in the epilogue (W = [gate; up], 2I rows):
//     y[i] = bf16(silu(W[i].x)) * (W[I+i].x)
// Replaces F.linear -> hipBLASLt for M = 1 (at bs = 1 the
// 15 GB of weights are the largest HBM term of a decode step; the library kernels run them at 2.2-4.7 TB/s).
//
// HBM-bound streaming: a wave owns R weight rows; each wave-instruction reads 1 KiB of a row (64 lanes x
// 16 B, coalesced), R x 4 row segments are in flight per wave before the first use, x (<= 28 KB) comes from
// L2.  Per 16 B of weights: 8 conversions + 4 packed FMAs, far below the VALU roof.  Rows are reduced over the
// 64 lanes with a DPP butterfly + two cross-row shuffles; lane 0 writes.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

// NORM variant (K == 4096): x is the pre-norm hidden state; every wave recomputes h = x + residual (bf16),
// the RMS statistics (its 64 lanes x 8 k-steps cover all 4096 elements) and xn = bf16(h * rstd * w_norm) in
// registers, so "residual add + RMSNorm + projection" is one launch; block 0 / wave 0 stores h (the new
// residual stream).  Same rounding points as the separate skv_add_rmsnorm launch.
template <int R, bool SILU_PAIR, bool NORM>
__global__ __launch_bounds__(256) void skv_gemv_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int N,
                                                       int K, int I /* SILU_PAIR: rows of one half */,
                                                       const bf16_t* __restrict__ residual,
                                                       const bf16_t* __restrict__ w_norm, bf16_t* __restrict__ h_out,
                                                       float eps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksteps = NORM ? 8 : K / 512;  // 64 lanes x 8 elements per step (NORM: K == 4096, static trip count)
    // rows of this wave
    int rows[R];
    const int unit0 = (blockIdx.x * 4 + wave) * (SILU_PAIR ? R / 2 : R);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (SILU_PAIR) rows[r] = (r & 1) ? I + unit0 + r / 2 : unit0 + r / 2;  // (gate, up) pairs
        else rows[r] = unit0 + r;
    }
    const int limit = SILU_PAIR ? I : N;
    if (unit0 >= limit) return;
    const bf16_t* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int row = rows[r];
        const int base_unit = SILU_PAIR ? unit0 + r / 2 : unit0 + r;
        if (base_unit >= limit) row = SILU_PAIR ? ((r & 1) ? I : 0) : 0;  // clamp: computed and discarded
        wp[r] = W + (size_t)row * K + 8 * lane;
    }
    const bf16_t* xp = x + 8 * lane;
    // the first weight segments do not depend on the norm prologue: get them in flight before it
    u32x4 wpre[NORM ? R : 1][4];
    if (NORM) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r)
                wpre[r][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + (size_t)u * 512));
    }
    f32x2 xn[NORM ? 8 : 1][4];
    if (NORM) {
        u32x4 hx[8];
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u32x4 a = *reinterpret_cast<const u32x4*>(xp + (size_t)u * 512);
            if (residual) {
                u32x4 c = *reinterpret_cast<const u32x4*>(residual + 8 * lane + (size_t)u * 512);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = pack_bf2(bf_lo(a[j]) + bf_lo(c[j]), bf_hi(a[j]) + bf_hi(c[j]));
            }
            hx[u] = a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ss = __builtin_fmaf(bf_lo(a[j]), bf_lo(a[j]), ss);
                ss = __builtin_fmaf(bf_hi(a[j]), bf_hi(a[j]), ss);
            }
        }
        if (h_out && blockIdx.x == 0 && wave == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<u32x4*>(h_out + 8 * lane + (size_t)u * 512) = hx[u];
        }
        ss = wave_tree_sum(ss);
        const float rstd = 1.0f / sqrtf(ss / (float)K + eps);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u32x4 g = *reinterpret_cast<const u32x4*>(w_norm + 8 * lane + (size_t)u * 512);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xn[u][j] = (f32x2){bfr(bf_lo(hx[u][j]) * rstd * bf_lo(g[j])), bfr(bf_hi(hx[u][j]) * rstd * bf_hi(g[j]))};
        }
    }
    f32x2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = (f32x2){0.f, 0.f};

#pragma unroll 2
    for (int ks = 0; ks < ksteps; ks += 4) {
        u32x4 wv[R][4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks + u < ksteps) {
                if (!NORM) xv[u] = *reinterpret_cast<const u32x4*>(xp + (size_t)(ks + u) * 512);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (NORM && ks == 0) wv[r][u] = wpre[r][u];
                    else wv[r][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + (size_t)(ks + u) * 512));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks + u < ksteps) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 xx = NORM ? xn[(ks + u) & 7][j] : (f32x2){bf_lo(xv[u][j]), bf_hi(xv[u][j])};
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const f32x2 ww = (f32x2){bf_lo(wv[r][u][j]), bf_hi(wv[r][u][j])};
                        acc[r][j] = __builtin_elementwise_fma(ww, xx, acc[r][j]);
                    }
                }
            }
        }
    }
    float tot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = ((acc[r][0].x + acc[r][0].y) + (acc[r][1].x + acc[r][1].y)) +
                  ((acc[r][2].x + acc[r][2].y) + (acc[r][3].x + acc[r][3].y));
        tot[r] = wave_tree_sum(s);
    }
    if (lane == 0) {
        if (SILU_PAIR) {
#pragma unroll
            for (int p = 0; p < R / 2; ++p) {
                const int i = unit0 + p;
                if (i < I) {
                    const float g = bfr(tot[2 * p]), u = bfr(tot[2 * p + 1]);   // the GEMV outputs are bf16 tensors
                    y[i] = f2bf(bfr(g / (1.0f + __expf(-g))) * u);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = unit0 + r;
                if (n < N) y[n] = f2bf(bias ? bfr(tot[r]) + bf2f(bias[n]) : tot[r]);
            }
        }
    }
}

static int launch_gemv(const void* W, const void* x, const void* bias, void* y, int N, int K, int fuse_silu_mul,
                       const void* residual, const void* w_norm, void* h_out, float eps, bool norm, hipStream_t st) {
    if (!W || !x || !y || N < 1) return SKV_ERR_ARG;
    if (K % 512 || K < 512) return SKV_ERR_UNSUPPORTED;
    if (norm && (K != 4096 || !w_norm)) return SKV_ERR_UNSUPPORTED;
#define SKV_GEMV(SILU, NORMF, GRID, IARG)                                                                            \
    hipLaunchKernelGGL((skv_gemv_kernel<4, SILU, NORMF>), dim3(GRID), dim3(256), 0, st, (const bf16_t*)W,             \
                       (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)y, N, K, IARG, (const bf16_t*)residual,        \
                       (const bf16_t*)w_norm, (bf16_t*)h_out, eps)
    if (fuse_silu_mul) {
        if (N % 2 || bias) return SKV_ERR_ARG;
        const int I = N / 2;
        const int grid = (I + 4 * 2 - 1) / (4 * 2);  // 4 waves x 2 (gate, up) pairs
        if (norm) SKV_GEMV(true, true, grid, I); else SKV_GEMV(true, false, grid, I);
    } else {
        const int grid = (N + 4 * 4 - 1) / (4 * 4);
        if (norm) SKV_GEMV(false, true, grid, 0); else SKV_GEMV(false, false, grid, 0);
    }
#undef SKV_GEMV
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

extern "C" int skv_gemv_bf16(const void* W, const void* x, const void* bias, void* y, int N, int K, int fuse_silu_mul,
                             skv_stream_t stream) {
    return launch_gemv(W, x, bias, y, N, K, fuse_silu_mul, nullptr, nullptr, nullptr, 0.f, false, (hipStream_t)stream);
}

extern "C" int skv_norm_gemv_bf16(const void* W, const void* x, const void* residual, const void* norm_weight, float eps,
                                  void* h_out, const void* bias, void* y, int N, int K, int fuse_silu_mul,
                                  skv_stream_t stream) {
    return launch_gemv(W, x, bias, y, N, K, fuse_silu_mul, residual, norm_weight, h_out, eps, true, (hipStream_t)stream);
}
