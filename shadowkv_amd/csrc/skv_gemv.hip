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

template <int R, bool SILU_PAIR>
__global__ __launch_bounds__(256) void skv_gemv_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int N,
                                                       int K, int I /* SILU_PAIR: rows of one half */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksteps = K / 512;  // 64 lanes x 8 elements per step
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
    f32x2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = (f32x2){0.f, 0.f};

    for (int ks = 0; ks < ksteps; ks += 4) {
        u32x4 wv[R][4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks + u < ksteps) {
                xv[u] = *reinterpret_cast<const u32x4*>(xp + (size_t)(ks + u) * 512);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    wv[r][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + (size_t)(ks + u) * 512));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks + u < ksteps) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 xx = (f32x2){bf_lo(xv[u][j]), bf_hi(xv[u][j])};
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

extern "C" int skv_gemv_bf16(const void* W, const void* x, const void* bias, void* y, int N, int K, int fuse_silu_mul,
                             skv_stream_t stream) {
    if (!W || !x || !y || N < 1) return SKV_ERR_ARG;
    if (K % 512 || K < 512) return SKV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (fuse_silu_mul) {
        if (N % 2 || bias) return SKV_ERR_ARG;
        const int I = N / 2;
        const int grid = (I + 4 * 2 - 1) / (4 * 2);  // 4 waves x 2 (gate, up) pairs
        hipLaunchKernelGGL((skv_gemv_kernel<4, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)W, (const bf16_t*)x,
                           (const bf16_t*)nullptr, (bf16_t*)y, N, K, I);
    } else {
        const int grid = (N + 4 * 4 - 1) / (4 * 4);
        hipLaunchKernelGGL((skv_gemv_kernel<4, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)W, (const bf16_t*)x,
                           (const bf16_t*)bias, (bf16_t*)y, N, K, 0);
    }
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}
