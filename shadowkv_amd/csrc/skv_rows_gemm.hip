// Dense projections of a decode step with a handful of token rows (bs = 2..32 sequences, q_len == 1):
//     Y[m][n] = sum_k W[n][k] * X[m][k] (+ bias[n]),   W [N][K] bf16 row-major, X [M][K], Y [M][N], M <= 32
// and the gate/up variant (W = [gate; up], 2I rows): Y[m][i] = bf16(silu(W[i].x_m)) * (W[I+i].x_m).
// SURVEY.md section 8f rank 3 (batched decode): one GEMV per token row streams the 15 GB of weights bs times; here
// they stream ONCE and the M tokens ride the N dimension of v_mfma_f32_16x16x32_bf16, so the multiply costs no VALU
// work at all (a VALU GEMV needs 4*M packed FMAs per 16 B of weights - past the issue budget from M ~ 8).
//
// HBM-bound streaming, same roof as skv_gemv.hip.  Workgroup = 8 waves on ONE tile of 16 weight rows, K split 8
// ways (so N = 4096 still launches 2,048 waves); each wave runs K/8 as MFMA steps of 32 k:
//     A operand (weights): lane (a = lane & 15, g = lane >> 4) loads 16 B = W[row a][32 s + 8 g .. +8]  (nontemporal;
//                          a wave-instruction covers 16 rows x 64 B, 8 unrolled steps 16 rows x 512 B in flight)
//     B operand (tokens) : lane (a, g) loads X[token min(a, M-1)][32 s + 8 g .. +8] from L2 (<= 16 x K x 2 B, shared
//                          by every wave of the launch); columns >= M are computed on a duplicate and never stored
//     D                  : lane holds D[rows 4 g .. 4 g + 3][token a]
// The 8 partial tiles meet in LDS and are summed in wave order (deterministic), then bias / SiLU*mul and one bf16
// rounding like the GEMV.  Accumulation order differs from the one-token GEMV (MFMA tree vs fma chain): results agree
// to f32 accumulation error, not bit for bit - the batched parity tests carry that tolerance.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifndef RG_WAVES
#define RG_WAVES 8
#endif
#ifndef RG_U
#define RG_U 8
#endif
// RG_WAVES: waves per 16-row tile (K split);  RG_U: MFMA steps (16-B loads per operand) in flight per wave

// PAIRX (M <= 8): token columns 8..15 of the MFMA are never stored, so their lanes (a >= 8) load the token operand of
// the NEXT k-step instead (token a - 8) and hand it over with one DPP row rotate: one token-operand load serves two
// MFMA steps (an ablation showed these L2 loads costing ~20 % of the kernel, profiles/r01_mfma_counters.txt).
// MT = 2 (17 <= M <= 32, the reference's batch 24 at 122K: test/e2e.py:63-68): two token tiles per weight load - the
// weight fragment of a k-step feeds two MFMAs (tokens 0..15 and 16..31), so the second half of the batch costs no
// extra weight traffic.
template <bool SILU_PAIR, bool PAIRX, int MT = 1>
__global__ __launch_bounds__(RG_WAVES * 64) void skv_rows_gemm_kernel(const bf16_t* __restrict__ W,
                                                                      const bf16_t* __restrict__ X,
                                                                      const bf16_t* __restrict__ bias,
                                                                      bf16_t* __restrict__ Y, int N, int K, int M,
                                                                      int I /* SILU_PAIR: rows of one half */) {
    static_assert(!(PAIRX && MT > 1), "the paired token operand serves one tile of <= 8 tokens");
    __shared__ float s_red[RG_WAVES][MT][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = lane & 15, g = lane >> 4;
    int row;
    if (SILU_PAIR) {  // tile = 8 gate rows (a < 8) + the matching 8 up rows
        int idx = blockIdx.x * 8 + (a & 7);
        if (idx >= I) idx = I - 1;  // clamp: computed and discarded
        row = a < 8 ? idx : I + idx;
    } else {
        row = blockIdx.x * 16 + a;
        if (row >= N) row = N - 1;
    }
    const bf16_t* wp = W + (size_t)row * K + 8 * g;
    const int tokl = PAIRX ? (a & 7) : a;
    const bf16_t* xp[MT];
    const bf16_t* xs[MT];                                               // single steps: every lane its own step
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int tok = min(16 * t + tokl, M - 1);                       // columns >= M: a duplicate, never stored
        xs[t] = X + (size_t)tok * K + 8 * g;
        xp[t] = xs[t] + (PAIRX ? (a >> 3) * 32 : 0);
    }
    const int S = K / 32, per = (S + RG_WAVES - 1) / RG_WAVES;
    const int s0 = wave * per, s1 = min(S, s0 + per);
    f32x4 acc0[MT], acc1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int s = s0;
    for (; s + RG_U <= s1; s += RG_U) {
        u32x4 wv[RG_U], xv[MT][RG_U];
#pragma unroll
        for (int u = 0; u < RG_U; ++u)
            wv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)(s + u) * 32));
        if (PAIRX) {
#pragma unroll
            for (int u = 0; u < RG_U; u += 2) {   // lanes a < 8: step s+u, lanes a >= 8: step s+u+1 (pointer offset above)
                xv[0][u] = *reinterpret_cast<const u32x4*>(xp[0] + (size_t)(s + u) * 32);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xv[0][u + 1][j] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xv[0][u][j], 0x128 /*row_ror:8*/, 0xf, 0xf, false);
            }
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < RG_U; ++u) xv[t][u] = *reinterpret_cast<const u32x4*>(xp[t] + (size_t)(s + u) * 32);
        }
#pragma unroll
        for (int u = 0; u < RG_U; u += 2) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[u]),
                                                                  __builtin_bit_cast(bf16x8, xv[t][u]), acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[u + 1]),
                                                                  __builtin_bit_cast(bf16x8, xv[t][u + 1]), acc1[t], 0, 0, 0);
            }
        }
    }
    for (; s < s1; ++s) {
        u32x4 wv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)s * 32));
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            u32x4 xv = *reinterpret_cast<const u32x4*>(xs[t] + (size_t)s * 32);
            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, xv),
                                                              acc0[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[wave][t][4 * g + r][a] = acc0[t][r] + acc1[t][r];
    __syncthreads();
    if (SILU_PAIR) {
        if (tid < 8 * M) {
            const int tok = tid >> 3, i = tid & 7, idx = blockIdx.x * 8 + i;
            if (idx < I) {
                float gs = 0.f, us = 0.f;
#pragma unroll
                for (int w = 0; w < RG_WAVES; ++w) {
                    gs += s_red[w][tok >> 4][i][tok & 15];
                    us += s_red[w][tok >> 4][i + 8][tok & 15];
                }
                const float gg = bfr(gs), uu = bfr(us);  // the projection outputs are bf16 tensors
                Y[(size_t)tok * I + idx] = f2bf(bfr(gg / (1.0f + __expf(-gg))) * uu);
            }
        }
    } else if (tid < 16 * M) {
        const int tok = tid >> 4, r = tid & 15, n = blockIdx.x * 16 + r;
        if (n < N) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < RG_WAVES; ++w) t += s_red[w][tok >> 4][r][tok & 15];
            Y[(size_t)tok * N + n] = f2bf(bias ? bfr(t) + bf2f(bias[n]) : t);
        }
    }
}

extern "C" int skv_linear_rows_bf16(const void* W, const void* X, const void* bias, void* Y, int M, int N, int K,
                                    int fuse_silu_mul, skv_stream_t stream) {
    if (!W || !X || !Y || M < 1 || N < 1) return SKV_ERR_ARG;
    if (M > 32 || K % 32 || K < 32) return SKV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (fuse_silu_mul) {
        if (N % 2 || bias) return SKV_ERR_ARG;
        const int I = N / 2;
        if (M <= 8)
            hipLaunchKernelGGL((skv_rows_gemm_kernel<true, true>), dim3((I + 7) / 8), dim3(RG_WAVES * 64), 0, st,
                               (const bf16_t*)W, (const bf16_t*)X, (const bf16_t*)nullptr, (bf16_t*)Y, N, K, M, I);
        else if (M > 16)
            hipLaunchKernelGGL((skv_rows_gemm_kernel<true, false, 2>), dim3((I + 7) / 8), dim3(RG_WAVES * 64), 0, st,
                               (const bf16_t*)W, (const bf16_t*)X, (const bf16_t*)nullptr, (bf16_t*)Y, N, K, M, I);
        else
            hipLaunchKernelGGL((skv_rows_gemm_kernel<true, false>), dim3((I + 7) / 8), dim3(RG_WAVES * 64), 0, st,
                               (const bf16_t*)W, (const bf16_t*)X, (const bf16_t*)nullptr, (bf16_t*)Y, N, K, M, I);
    } else {
        if (M <= 8)
            hipLaunchKernelGGL((skv_rows_gemm_kernel<false, true>), dim3((N + 15) / 16), dim3(RG_WAVES * 64), 0, st,
                               (const bf16_t*)W, (const bf16_t*)X, (const bf16_t*)bias, (bf16_t*)Y, N, K, M, 0);
        else if (M > 16)
            hipLaunchKernelGGL((skv_rows_gemm_kernel<false, false, 2>), dim3((N + 15) / 16), dim3(RG_WAVES * 64), 0, st,
                               (const bf16_t*)W, (const bf16_t*)X, (const bf16_t*)bias, (bf16_t*)Y, N, K, M, 0);
        else
            hipLaunchKernelGGL((skv_rows_gemm_kernel<false, false>), dim3((N + 15) / 16), dim3(RG_WAVES * 64), 0, st,
                               (const bf16_t*)W, (const bf16_t*)X, (const bf16_t*)bias, (bf16_t*)Y, N, K, M, 0);
    }
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}
