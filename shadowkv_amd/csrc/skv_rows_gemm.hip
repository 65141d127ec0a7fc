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

// Second-generation kernel (all M): a wave walks RT tiles of 16 weight rows over its K slice, so ONE token-operand load
// per k-step serves RT (x MT) MFMAs.  The first kernel above reloads the token operand (from L2 / L1) for every weight
// fragment - at MT = 2 two token loads per weight load, three times the HBM bytes through the CU's L1 - and ran the
// gate/up projection of 24 token rows at 2.8 TB/s (rocprof, bs 24); here the ratio is MT / RT.
//   RT x MT independent accumulators give the MFMA pipeline its parallelism (no second accumulator set);
//   U = 4 k-steps in flight: RT * U weight loads of 1 KB per wave.
template <bool SILU_PAIR, int MT, int RT, int NWV, int U = 4>
__global__ __launch_bounds__(NWV * 64) void skv_rows_gemm2_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X,
                                                                  const bf16_t* __restrict__ bias, bf16_t* __restrict__ Y,
                                                                  int N, int K, int M, int I /* SILU_PAIR: rows of one half */) {
    __shared__ float s_red[NWV][MT][RT][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = lane & 15, g = lane >> 4;
    const bf16_t* wp[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        int row;
        if (SILU_PAIR) {  // tile = 8 gate rows (a < 8) + the matching 8 up rows
            int idx = (blockIdx.x * RT + r) * 8 + (a & 7);
            if (idx >= I) idx = I - 1;  // clamp: computed and discarded
            row = a < 8 ? idx : I + idx;
        } else {
            row = (blockIdx.x * RT + r) * 16 + a;
            if (row >= N) row = N - 1;
        }
        wp[r] = W + (size_t)row * K + 8 * g;
    }
    const bf16_t* xs[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xs[t] = X + (size_t)min(16 * t + a, M - 1) * K + 8 * g;   // columns >= M: a duplicate, never stored
    const int S = K / 32, per = (S + NWV - 1) / NWV;
    const int s0 = wave * per, s1 = min(S, s0 + per);
    f32x4 acc[RT][MT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int s = s0;
    for (; s + U <= s1; s += U) {
        u32x4 wv[RT][U], xv[MT][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
                wv[r][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + (size_t)(s + u) * 32));
#pragma unroll
            for (int t = 0; t < MT; ++t) xv[t][u] = *reinterpret_cast<const u32x4*>(xs[t] + (size_t)(s + u) * 32);
        }
        // every load of the batch is issued before the first MFMA (hipcc otherwise sinks each load to its use and the
        // wave has two or three loads in flight instead of RT * U + MT * U)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[r][u]),
                                                                        __builtin_bit_cast(bf16x8, xv[t][u]), acc[r][t], 0, 0, 0);
    }
    for (; s < s1; ++s) {
        u32x4 xv[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) xv[t] = *reinterpret_cast<const u32x4*>(xs[t] + (size_t)s * 32);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const u32x4 wv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + (size_t)s * 32));
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, xv[t]),
                                                                    acc[r][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) s_red[wave][t][r][4 * g + i][a] = acc[r][t][i];
    __syncthreads();
    if (SILU_PAIR) {
        for (int o = tid; o < RT * 8 * M; o += NWV * 64) {
            const int tok = o / (RT * 8), j = o % (RT * 8), r = j >> 3, i = j & 7, idx = (blockIdx.x * RT + r) * 8 + i;
            if (idx < I) {
                float gs = 0.f, us = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) {
                    gs += s_red[w][tok >> 4][r][i][tok & 15];
                    us += s_red[w][tok >> 4][r][i + 8][tok & 15];
                }
                const float gg = bfr(gs), uu = bfr(us);  // the projection outputs are bf16 tensors
                Y[(size_t)tok * I + idx] = f2bf(bfr(gg / (1.0f + __expf(-gg))) * uu);
            }
        }
    } else {
        for (int o = tid; o < RT * 16 * M; o += NWV * 64) {
            const int tok = o / (RT * 16), j = o % (RT * 16), r = j >> 4, rr = j & 15, n = (blockIdx.x * RT + r) * 16 + rr;
            if (n < N) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) t += s_red[w][tok >> 4][r][rr][tok & 15];
                Y[(size_t)tok * N + n] = f2bf(bias ? bfr(t) + bf2f(bias[n]) : t);
            }
        }
    }
}

template <bool SILU_PAIR, int MT, int RT, int NWV, int U = 4>
static void launch_rows2(const void* W, const void* X, const void* bias, void* Y, int M, int N, int K, int I, hipStream_t st) {
    const int tiles = SILU_PAIR ? (I + 7) / 8 : (N + 15) / 16;
    hipLaunchKernelGGL((skv_rows_gemm2_kernel<SILU_PAIR, MT, RT, NWV, U>), dim3((tiles + RT - 1) / RT), dim3(NWV * 64), 0, st,
                       (const bf16_t*)W, (const bf16_t*)X, (const bf16_t*)bias, (bf16_t*)Y, N, K, M, I);
}

template <bool SILU_PAIR, bool PAIRX, int MT>
static void launch_rows1(const void* W, const void* X, const void* bias, void* Y, int M, int N, int K, int I, hipStream_t st) {
    hipLaunchKernelGGL((skv_rows_gemm_kernel<SILU_PAIR, PAIRX, MT>), dim3(SILU_PAIR ? (I + 7) / 8 : (N + 15) / 16),
                       dim3(RG_WAVES * 64), 0, st, (const bf16_t*)W, (const bf16_t*)X, (const bf16_t*)bias, (bf16_t*)Y, N, K, M, I);
}

// Kernel and shape per (token rows, projection), measured on MI355X with HBM-cold weights (tools/time_rows_gemm.py, us;
// g1 = first-generation kernel, RTxNW = second generation with RT row tiles per wave and NW waves per workgroup):
//     rows   QKV 6144x4096        O 4096x4096        gate/up 28672x4096        down 4096x14336
//      8     g1 15.6 | 2x8 15.2   g1  9.3 | 1x8 10.1  g1 52.3 | 4x4 59.9        g1 29.4 | 1x8 32.0      -> g1 (paired operand)
//     16     g1 18.9 | 2x8 16.6   g1 10.6 | 1x8 11.8  g1 63.7 | 4x4 62.9        g1 33.5 | 1x8 36.7      -> g1, QKV 2x8
//     24     g1 25.1 | 2x8 19.4   g1 13.6 | 1x8 13.9  g1 85.6 | 4x4 59.9        g1 45.6 | 1x8 42.8      -> 2x8, g1, 4x4, 1x8
//     32     g1 28.6 | 2x8 20.7   g1 15.6 | 1x8 15.0  g1 96.5 | 4x4 62.3        g1 50.6 | 1x8 47.4
// (fewer, fatter workgroups lose where the projection has only 256 row tiles; the token operand's distinct cache lines,
// not the weight stream, set the pace of the first generation from 17 rows on.)
template <bool SILU_PAIR>
static void launch_rows(const void* W, const void* X, const void* bias, void* Y, int M, int N, int K, int I, hipStream_t st) {
    const bool mid = N >= 6144 && N < 16384;
    if (M <= 8) return launch_rows1<SILU_PAIR, true, 1>(W, X, bias, Y, M, N, K, I, st);
    if (M <= 16) {
        if (mid) return launch_rows2<SILU_PAIR, 1, 2, 8>(W, X, bias, Y, M, N, K, I, st);
        return launch_rows1<SILU_PAIR, false, 1>(W, X, bias, Y, M, N, K, I, st);
    }
    if (N >= 16384) return launch_rows2<SILU_PAIR, 2, 4, 4>(W, X, bias, Y, M, N, K, I, st);
    if (mid) return launch_rows2<SILU_PAIR, 2, 2, 8>(W, X, bias, Y, M, N, K, I, st);
    if (K >= 8192) return launch_rows2<SILU_PAIR, 2, 1, 8>(W, X, bias, Y, M, N, K, I, st);
    return launch_rows1<SILU_PAIR, false, 2>(W, X, bias, Y, M, N, K, I, st);
}

extern "C" int skv_linear_rows_bf16(const void* W, const void* X, const void* bias, void* Y, int M, int N, int K,
                                    int fuse_silu_mul, skv_stream_t stream) {
    if (!W || !X || !Y || M < 1 || N < 1) return SKV_ERR_ARG;
    if (M > 32 || K % 32 || K < 32) return SKV_ERR_UNSUPPORTED;
    if (fuse_silu_mul && (N % 2 || bias)) return SKV_ERR_ARG;
    if (fuse_silu_mul) launch_rows<true>(W, X, nullptr, Y, M, N, K, N / 2, (hipStream_t)stream);
    else launch_rows<false>(W, X, bias, Y, M, N, K, 0, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}
