// bf16 GEMV for the decode step's dense projections (q_len == 1, one token per sequence):
//     y[n] = sum_k W[n][k] * x[k] (+ bias[n]),   W [N][K] row-major (K contiguous), f32 accumulation, bf16 out
// and the gate/up variant with the activation fused in the epilogue (W = [gate; up], 2I rows):
//     y[i] = bf16(silu(W[i].x)) * (W[I+i].x)
// Replaces F.linear -> hipBLASLt for M = 1 (at bs = 1 the
// 15 GB of weights are the largest HBM term of a decode step; the library kernels run them at 2.2-4.7 TB/s).
//
// K % 8 == 0, K >= 512 (a partial last 512-element step is handled after the main loop).
// HBM-bound streaming: a wave owns R weight rows; each wave-instruction reads 1 KiB of a row (64 lanes x
// 16 B, coalesced), R x 4 row segments are in flight per wave before the first use, x (<= 28 KB) comes from
// L2.  Per 16 B of weights: 8 conversions + 4 packed FMAs, far below the VALU roof.  Rows are reduced over the
// 64 lanes with a DPP butterfly + two cross-row shuffles; lane 0 writes.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"
#include "skv_early.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef SKV_GEMV_PRE_R2
#define SKV_GEMV_PRE_R2 4      // k-steps of weights a two-row wave of the NORM variant requests before its norm prologue (round 5: all 8
                               // up front - 142 VGPRs - measured 228.8-230.1 tokens/s against 230.3-231.3 with 4 on one box: not kept)
#endif

// NORM variant (K == 4096): x is the pre-norm hidden state; every BLOCK recomputes h = x + residual (bf16), the
// RMS statistics and xn = bf16(h * rstd * w_norm) cooperatively (256 threads x 16 elements, through LDS), so
// "residual add + RMSNorm + projection" is one launch; block 0 stores h (the new residual stream).  Same
// element -> thread mapping and reduction tree as skv_add_rmsnorm_kernel: bit-identical to the separate launch.
// QKV epilogue (QKV == true, W = fused [q; k; v] projection of ONE token, head_dim 128): the wave's 4 rows are two
// rotation pairs of one head - NeoX: (t, t+64), (t+1, t+65); GLM: (2t', 2t'+1), (2t'+2, 2t'+3) - so lane 0 can rotate
// q / k at the token's position and write q to q_out and k / v straight into the cache row: the projection, the
// split, RoPE and update_kv_cache are one launch (same arithmetic as skv_qkv_rope_update_kernel).
struct QkvEpilogue {
    const bf16_t* cos_sin;
    const int64_t* pos;       // [1]
    const int64_t* row_idx;   // [1]
    const bf16_t* q_override; // nullable [Hq][128]
    bf16_t* q_out;            // [Hq][128]
    bf16_t* k_cache;          // [Hkv][cache_rows][128]
    bf16_t* v_cache;
    long long cs_stride, cache_stride_h;
    int Hq, Hkv, cache_rows, glm, cs_rows;
};

// RMAX (lm_head, N % 16 == 0): the workgroup also leaves the LARGEST of its 16 outputs as an order-preserving 16-bit key in
// range_max[blockIdx.x] - the sampler (skv_sample.hip) then finds the rows that can hold a top-k logit from N / 16 keys
// instead of streaming all N logits through one CU.
// NEARP (round 5, the gate/up launch of a ShadowKV layer): the first np.blocks workgroups do not compute - they stage the
// near misses of this step's selection ahead of the next step (skv_near_pull_role, skv_early.h) while the GEMV streams; the
// GEMV's own workgroups follow them in the grid.
template <int R, bool SILU_PAIR, bool NORM, bool QKV, bool RMAX = false, bool NEARP = false>
__global__ __launch_bounds__(256) void skv_gemv_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int N,
                                                       int K, int I /* SILU_PAIR: rows of one half */,
                                                       const bf16_t* __restrict__ residual,
                                                       const bf16_t* __restrict__ w_norm, bf16_t* __restrict__ h_out,
                                                       float eps, QkvEpilogue qe, uint16_t* __restrict__ range_max = nullptr,
                                                       NearPull np = NearPull{}) {
    int bx = blockIdx.x;
    if constexpr (NEARP) {
        if (bx < np.blocks) {
            __shared__ int s_near[6 * 64 + 8];
            skv_near_pull_role(np, bx, threadIdx.x, s_near);
            return;
        }
        bx -= np.blocks;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksteps = NORM ? 8 : K / 512;  // 64 lanes x 8 elements per step (NORM: K == 4096, static trip count)
    // rows of this wave
    int rows[R];
    const int unit0 = (bx * 4 + wave) * (SILU_PAIR ? R / 2 : R);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (SILU_PAIR) rows[r] = (r & 1) ? I + unit0 + r / 2 : unit0 + r / 2;  // (gate, up) pairs
        else if (QKV && !qe.glm) {
            // unit0 = R * wave index; a wave owns R/2 rotation pairs (t, t+64) of one head: 128/R waves per head,
            // rows t0, t0+64, t0+1, t0+65, ...
            const int w = unit0 / R, head = w / (128 / R), t0 = (R / 2) * (w % (128 / R));
            rows[r] = head * 128 + t0 + (r >> 1) + ((r & 1) ? 64 : 0);
        } else rows[r] = unit0 + r;
    }
    const int limit = SILU_PAIR ? I : N;
    if (!NORM && unit0 >= limit) return;   // (NORM: every wave takes part in the block-wide prologue first)
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
    constexpr int PRE = (NORM && R <= 2) ? SKV_GEMV_PRE_R2 : 4;
    u32x4 wpre[NORM ? R : 1][PRE];
    if (NORM) {
#pragma unroll
        for (int u = 0; u < PRE; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r)
                wpre[r][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + (size_t)u * 512));
    }
    f32x2 xn[NORM ? 8 : 1][4];
    if (NORM) {
        // Block-cooperative prologue: the 256 threads split the 4096 elements exactly like skv_add_rmsnorm_kernel
        // (thread t owns 16-B vectors t and t + 256, same per-thread order, same wave tree, same (s0+s1)+(s2+s3)),
        // so h, rstd and xn are bit-identical to the separate launch; xn goes through LDS and every wave picks up
        // its k-step slices (vector 64 u + lane for step u).  One read of x / residual / w_norm per BLOCK.
        __shared__ u32x4 s_xn[512];
        __shared__ float s_ss[4];
        const int tid = threadIdx.x;
        u32x4 hx[2];
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int v = tid + it * 256;
            u32x4 a = reinterpret_cast<const u32x4*>(x)[v];
            if (residual) {
                u32x4 c = reinterpret_cast<const u32x4*>(residual)[v];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = pack_bf2(bf_lo(a[j]) + bf_lo(c[j]), bf_hi(a[j]) + bf_hi(c[j]));
            }
            hx[it] = a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ss = __builtin_fmaf(bf_lo(a[j]), bf_lo(a[j]), ss);
                ss = __builtin_fmaf(bf_hi(a[j]), bf_hi(a[j]), ss);
            }
        }
        if (h_out && bx == 0) {
#pragma unroll
            for (int it = 0; it < 2; ++it) reinterpret_cast<u32x4*>(h_out)[tid + it * 256] = hx[it];
        }
        ss = wave_tree_sum(ss);
        if (lane == 0) s_ss[wave] = ss;
        __syncthreads();
        const float tot = (s_ss[0] + s_ss[1]) + (s_ss[2] + s_ss[3]);
        const float rstd = 1.0f / sqrtf(tot / (float)K + eps);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int v = tid + it * 256;
            u32x4 g = reinterpret_cast<const u32x4*>(w_norm)[v];
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf2(bf_lo(hx[it][j]) * rstd * bf_lo(g[j]), bf_hi(hx[it][j]) * rstd * bf_hi(g[j]));
            s_xn[v] = o;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x4 o = s_xn[u * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) xn[u][j] = (f32x2){bf_lo(o[j]), bf_hi(o[j])};
        }
        if (unit0 >= limit) return;
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
                    if (NORM && ks + u < PRE) wv[r][u] = wpre[r][ks + u];
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
    if (!NORM) {   // K % 512 != 0 (GLM-4: 13696 = 26 x 512 + 384): the last partial step, lanes below the tail only
        const int tail_lanes = (K % 512) / 8;
        if (lane < tail_lanes) {
            const u32x4 xt = *reinterpret_cast<const u32x4*>(xp + (size_t)ksteps * 512);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const u32x4 wt = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + (size_t)ksteps * 512));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[r][j] = __builtin_elementwise_fma((f32x2){bf_lo(wt[j]), bf_hi(wt[j])},
                                                          (f32x2){bf_lo(xt[j]), bf_hi(xt[j])}, acc[r][j]);
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
    if (QKV) {
        if (lane == 0 && unit0 < N) {
            const int head = unit0 / 128;                       // the wave's R rows belong to one head
            float o[R];
#pragma unroll
            for (int r = 0; r < R; ++r) o[r] = bfr(bias ? bfr(tot[r]) + bf2f(bias[rows[r]]) : tot[r]);  // the projection output is bf16
            const long long row = *qe.row_idx;
            if (head >= qe.Hq + qe.Hkv) {                       // V: copy into the cache row
                if (row >= 0 && row < qe.cache_rows) {
                    bf16_t* dst = qe.v_cache + (long long)(head - qe.Hq - qe.Hkv) * qe.cache_stride_h + row * 128;
#pragma unroll
                    for (int r = 0; r < R; ++r) dst[rows[r] - head * 128] = f2bf(o[r]);
                }
            } else {
                const long long pcl = min(max(*qe.pos, 0ll), (long long)qe.cs_rows - 1);   // see skv_dense.hip
                const bf16_t* cs = qe.cos_sin + pcl * qe.cs_stride;
                float res[R];
                int dim[R];
#pragma unroll
                for (int p = 0; p < R / 2; ++p) {
                    const int d1 = rows[2 * p] - head * 128, d2 = rows[2 * p + 1] - head * 128;
                    dim[2 * p] = d1; dim[2 * p + 1] = d2;
                    const float x1 = o[2 * p], x2 = o[2 * p + 1];
                    if (!qe.glm) {                              // pair (t, t+64)
                        const float c = bf2f(cs[d1]), sn = bf2f(cs[d1 + 64]);
                        res[2 * p] = bfr(x1 * c) + bfr(-x2 * sn);
                        res[2 * p + 1] = bfr(x2 * c) + bfr(x1 * sn);
                    } else if (d1 < 64) {                       // pair (2t, 2t+1), cos[t], sin[32+t]
                        const float c = bf2f(cs[d1 >> 1]), sn = bf2f(cs[32 + (d1 >> 1)]);
                        res[2 * p] = bfr(x1 * c) + bfr(-x2 * sn);
                        res[2 * p + 1] = bfr(x2 * c) + bfr(x1 * sn);
                    } else {
                        res[2 * p] = x1; res[2 * p + 1] = x2;   // pass-through dims
                    }
                }
                if (head < qe.Hq) {
                    bf16_t* dst = qe.q_out + (size_t)head * 128;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float v = res[r];
                        if (qe.q_override) v = bf2f(qe.q_override[(size_t)head * 128 + dim[r]]) + v * 0.0f;
                        dst[dim[r]] = f2bf(v);
                    }
                } else if (row >= 0 && row < qe.cache_rows) {
                    bf16_t* dst = qe.k_cache + (long long)(head - qe.Hq) * qe.cache_stride_h + row * 128;
#pragma unroll
                    for (int r = 0; r < R; ++r) dst[dim[r]] = f2bf(res[r]);
                }
            }
        }
        return;
    }
    __shared__ uint32_t s_kmax[RMAX ? 4 : 1];
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
            uint32_t kmax = 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = unit0 + r;
                const bf16_t o = f2bf(bias ? bfr(tot[r]) + bf2f(bias[n < N ? n : 0]) : tot[r]);
                if (n < N) y[n] = o;
                if (RMAX) kmax = max(kmax, skv_bf16_order_key(o));
            }
            if constexpr (RMAX) s_kmax[wave] = kmax;
        }
    }
    if constexpr (RMAX && !SILU_PAIR && !QKV) {      // (N % 16 == 0: every wave of every workgroup arrives here)
        __syncthreads();
        if (tid == 0) range_max[bx] = (uint16_t)max(max(s_kmax[0], s_kmax[1]), max(s_kmax[2], s_kmax[3]));
    }
}

static int launch_gemv(const void* W, const void* x, const void* bias, void* y, int N, int K, int fuse_silu_mul,
                       const void* residual, const void* w_norm, void* h_out, float eps, bool norm, hipStream_t st,
                       const QkvEpilogue* qkv = nullptr, uint16_t* range_max = nullptr, const NearPull* near = nullptr) {
    if (!W || !x || (!y && !qkv) || N < 1) return SKV_ERR_ARG;
    QkvEpilogue qe{};
    if (qkv) qe = *qkv;
    if (K % 8 || K < 512) return SKV_ERR_UNSUPPORTED;
    if (norm && (K != 4096 || !w_norm)) return SKV_ERR_UNSUPPORTED;
    if (near) {          // a launch of a layer's dense tail with the near-miss pull role in front of its grid
        if (qkv || range_max || near->blocks < 1 || near->E + SKV_NEAR_SLOTS > 32767) return SKV_ERR_UNSUPPORTED;
        if (norm && fuse_silu_mul && !(N % 2) && !bias) {                    // gate/up: list 0
            const int I = N / 2, grid = (I + 7) / 8;
            hipLaunchKernelGGL((skv_gemv_kernel<4, true, true, false, false, true>), dim3(near->blocks + grid), dim3(256), 0, st,
                               (const bf16_t*)W, (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)y, N, K, I, (const bf16_t*)residual,
                               (const bf16_t*)w_norm, (bf16_t*)h_out, eps, qe, (uint16_t*)nullptr, *near);
        } else if (!norm && !fuse_silu_mul && N <= 8192) {                   // down projection (two rows per wave): list 1
            hipLaunchKernelGGL((skv_gemv_kernel<2, false, false, false, false, true>), dim3(near->blocks + (N + 7) / 8), dim3(256), 0, st,
                               (const bf16_t*)W, (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)y, N, K, 0, (const bf16_t*)residual,
                               (const bf16_t*)w_norm, (bf16_t*)h_out, eps, qe, (uint16_t*)nullptr, *near);
        } else {
            return SKV_ERR_UNSUPPORTED;
        }
        return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
    }
    if (range_max) {     // the lm_head with the sampler's range keys: norm prologue, whole workgroups of 16 rows
        if (!norm || qkv || fuse_silu_mul || N % 16) return SKV_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((skv_gemv_kernel<4, false, true, false, true>), dim3(N / 16), dim3(256), 0, st, (const bf16_t*)W,
                           (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)y, N, K, 0, (const bf16_t*)residual,
                           (const bf16_t*)w_norm, (bf16_t*)h_out, eps, qe, range_max);
        return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
    }
#define SKV_GEMV(SILU, NORMF, GRID, IARG)                                                                            \
    hipLaunchKernelGGL((skv_gemv_kernel<4, SILU, NORMF, false>), dim3(GRID), dim3(256), 0, st, (const bf16_t*)W,      \
                       (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)y, N, K, IARG, (const bf16_t*)residual,        \
                       (const bf16_t*)w_norm, (bf16_t*)h_out, eps, qe)
    if (qkv) {
        if (fuse_silu_mul || N != (qe.Hq + 2 * qe.Hkv) * 128) return SKV_ERR_ARG;
        const int grid = (N + 15) / 16;
        if (norm)   // one rotation pair per wave: 3,072 waves for N = 6144 (two pairs per wave: 200.4 -> 202.3 tok/s)
            hipLaunchKernelGGL((skv_gemv_kernel<2, false, true, true>), dim3((N + 7) / 8), dim3(256), 0, st, (const bf16_t*)W,
                               (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)nullptr, N, K, 0, (const bf16_t*)residual,
                               (const bf16_t*)w_norm, (bf16_t*)h_out, eps, qe);
        else
            hipLaunchKernelGGL((skv_gemv_kernel<4, false, false, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)W,
                               (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)nullptr, N, K, 0, (const bf16_t*)residual,
                               (const bf16_t*)w_norm, (bf16_t*)h_out, eps, qe);
        return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
    }
    if (fuse_silu_mul) {
        if (N % 2 || bias) return SKV_ERR_ARG;
        const int I = N / 2;
        const int grid = (I + 4 * 2 - 1) / (4 * 2);  // 4 waves x 2 (gate, up) pairs
        if (norm) SKV_GEMV(true, true, grid, I); else SKV_GEMV(true, false, grid, I);
    } else {
        const int grid = (N + 4 * 4 - 1) / (4 * 4);
        if (!norm && N <= 8192) {
            // few output rows (O and down projections: N = 4096 -> 256 blocks = 1 per CU with 4 rows per wave): 2 rows
            // per wave doubles the waves and the bytes in flight per CU (measured 198.3 -> 200.7 tok/s; 1 row: 199.3)
            hipLaunchKernelGGL((skv_gemv_kernel<2, false, false, false>), dim3((N + 7) / 8), dim3(256), 0, st,
                               (const bf16_t*)W, (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)y, N, K, 0,
                               (const bf16_t*)residual, (const bf16_t*)w_norm, (bf16_t*)h_out, eps, qe);
        } else if (norm) SKV_GEMV(false, true, grid, 0); else SKV_GEMV(false, false, grid, 0);
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

static int near_pull_of(NearPull& np, void* early_state, int blocks, int groups, int n_landmarks, int n_chunks, int early_max,
                        const void* v_host, long long host_block_stride, int pull_parts, int list, int active_lists) {
    if (pull_parts != 1 && pull_parts != 2 && pull_parts != 4) return SKV_ERR_ARG;
    if (list < 0 || list >= SKV_NEAR_LISTS || active_lists < 1 || active_lists > SKV_NEAR_LISTS || list >= active_lists) return SKV_ERR_ARG;
    if (!early_state || !v_host || blocks < 1 || groups < 1 || n_landmarks < 1 || n_chunks < 1 || early_max < 1 || early_max > 128 ||
        (host_block_stride % 8))
        return SKV_ERR_ARG;
    if (n_chunks > (1 << 18)) return SKV_ERR_UNSUPPORTED;      // (the early state's limit, skv_select_chunks_fused)
    const EarlyState es = skv_carve_early(early_state, blocks, groups, n_landmarks, n_chunks, early_max);
    np = skv_near_pull(es, v_host, host_block_stride, blocks, n_chunks, early_max, pull_parts, list, active_lists);
    return SKV_OK;
}

extern "C" int skv_norm_gemv_near_pull_bf16(const void* W, const void* x, const void* residual, const void* norm_weight, float eps,
                                            void* h_out, void* y, int N, int K, void* early_state, int blocks, int groups,
                                            int n_landmarks, int n_chunks, int early_max, const void* v_host,
                                            long long host_block_stride, int pull_parts, int active_lists, skv_stream_t stream) {
    NearPull np{};
    const int rc = near_pull_of(np, early_state, blocks, groups, n_landmarks, n_chunks, early_max, v_host, host_block_stride, pull_parts, 0,
                                active_lists);
    if (rc != SKV_OK) return rc;
    return launch_gemv(W, x, nullptr, y, N, K, 1, residual, norm_weight, h_out, eps, true, (hipStream_t)stream, nullptr, nullptr, &np);
}

// the down projection (skv_gemv_bf16, N <= 8192) with the pull role of near-miss list `list` (1: ranks S + 65 .. S + 128)
extern "C" int skv_gemv_near_pull_bf16(const void* W, const void* x, const void* bias, void* y, int N, int K, void* early_state,
                                       int blocks, int groups, int n_landmarks, int n_chunks, int early_max, const void* v_host,
                                       long long host_block_stride, int pull_parts, int list, skv_stream_t stream) {
    NearPull np{};
    const int rc = near_pull_of(np, early_state, blocks, groups, n_landmarks, n_chunks, early_max, v_host, host_block_stride, pull_parts, list,
                                SKV_NEAR_LISTS);
    if (rc != SKV_OK) return rc;
    return launch_gemv(W, x, bias, y, N, K, 0, nullptr, nullptr, nullptr, 0.f, false, (hipStream_t)stream, nullptr, nullptr, &np);
}

extern "C" int skv_norm_gemv_rangemax_bf16(const void* W, const void* x, const void* residual, const void* norm_weight, float eps,
                                           void* h_out, const void* bias, void* y, int N, int K, void* range_max,
                                           skv_stream_t stream) {
    if (!range_max) return SKV_ERR_ARG;
    return launch_gemv(W, x, bias, y, N, K, 0, residual, norm_weight, h_out, eps, true, (hipStream_t)stream, nullptr,
                       (uint16_t*)range_max);
}

extern "C" int skv_qkv_gemv_rope_update(const void* Wqkv, const void* x, const void* residual, const void* norm_weight,
                                        float eps, void* h_out, const void* bias, const void* cos_sin, const int64_t* pos,
                                        const int64_t* row_idx, const void* q_override, void* q_out, void* k_cache,
                                        void* v_cache, int K, int q_heads, int kv_heads, int head_dim,
                                        long long cos_sin_stride, int cos_sin_rows, long long cache_stride_h, int cache_rows,
                                        int rope_mode, skv_stream_t stream) {
    if (!cos_sin || !pos || !row_idx || !q_out || !k_cache || !v_cache || cos_sin_rows < 1) return SKV_ERR_ARG;
    if (head_dim != 128 || (rope_mode != 1 && rope_mode != 2)) return SKV_ERR_UNSUPPORTED;
    QkvEpilogue qe{(const bf16_t*)cos_sin, pos, row_idx, (const bf16_t*)q_override, (bf16_t*)q_out, (bf16_t*)k_cache,
                   (bf16_t*)v_cache, cos_sin_stride, cache_stride_h, q_heads, kv_heads, cache_rows, rope_mode == 2, cos_sin_rows};
    return launch_gemv(Wqkv, x, bias, nullptr, (q_heads + 2 * kv_heads) * 128, K, 0, residual, norm_weight, h_out, eps,
                       norm_weight != nullptr, (hipStream_t)stream, &qe);
}
