// Chunk selection for one decode step:
//   skv_score_tile_kernel        q . landmarks^T  -> bf16 logits + per-256-column partial (max, sum)
//   skv_softmax_final_kernel     partials -> (max, 1/sum) per row              [legacy 3-stage API]
//   skv_softmax_apply_kernel     P = bf16(exp(D-max)*inv)                      [legacy 3-stage API]
//   skv_normalize_groupmax_kernel  partials -> P -> max over the GQA group -> bf16 score per landmark
//   skv_topk_reorder_kernel      exact top-k (ties -> lowest slot), slot->chunk id, hit/miss diff
//                                against the resident chunk set, two sorts, offsets, cnts
//
// Replaces /root/reference/kernels/batch_gemm_softmax.{cu,h} (CUTLASS), the torch.max /
// torch.topk / gather chain at /root/reference/models/kv_cache.py:1023-1042 and
// /root/reference/kernels/map.cuh:754-796.  Arithmetic contract: skv_common.h ==
// oracle/shadowkv_oracle.c.
//
// Roofline: the scoring kernel streams the landmark table once (B*N*256 bytes, 31.9 MB per
// layer at the headline config) and is HBM-bound; everything after it touches <= 1.3 MB.
#include <stdlib.h>
#include "skv_common.h"
#include "skv_select_front.h"
#include "skv_launch.h"
#include "skv_early.h"

#define SKV_TILE 256  // columns per partial tile == the reference's ThreadblockShape::kN
// per-wave phase stamps of the scan (tools/score_probe.hip builds with -DSKV_SCORE_STAMPS; never in the library):
// g_score_stamps[(workgroup * 16 + wave) * 8 + i], 100 MHz wall clock, lane 0 of every wave
#ifdef SKV_SCORE_STAMPS
__device__ unsigned long long* g_score_stamps;
#define SCORE_STAMP(i)                                                                                                      \
    do {                                                                                                                    \
        if (g_score_stamps != nullptr && (threadIdx.x & 63) == 0)                                                           \
            g_score_stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define SCORE_STAMP(i)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// K1: scoring.  One workgroup (4 waves) per 256-landmark tile.  A wave-instruction loads
// 4 landmark rows (16 lanes x 16 B each, 1 KiB coalesced); lane `sub` of a row holds the
// 8 contiguous elements k = 8*sub..8*sub+7, multiplies them into G running sums (q lives
// in registers, two query heads per v_pk_fma_f32), and a 4-step DPP butterfly over the 16
// lanes finishes the dot products; the butterfly transposes (each lane finishes ONE of the G
// totals), so scaling, bf16 rounding and the LDS store happen once per lane, not G times.
// A tile is shared by 16 waves (16 rows = 4 loads each, all issued before the first use): many waves per SIMD
// keep VALU issue dense while the tile streams in (with 4 waves x 16 loads: 2.85 TB/s).
// ---------------------------------------------------------------------------------------
#ifndef SKV_SCORE_WAVES
#define SKV_SCORE_WAVES 16                      // waves per 256-landmark tile (16 rows each; 4/8/16 measured: 9.3 / 8.8 / 8.5 us)
#endif
// waves per SIMD the register allocation aims at: two workgroups per CU (8 per SIMD at 16 waves per tile and G <= 4: 64 VGPRs; 4 at
// 8 waves: 128; 2 at 4 waves: 256 - a wave that holds 16 row groups in flight needs them)
#define SKV_SCORE_MIN_WAVES(G, WAVES) ((WAVES) >= 16 ? ((G) >= 8 ? 4 : 8) : (WAVES) >= 8 ? 4 : (WAVES) >= 4 ? 2 : 1)
#ifndef SKV_SCORE_PD
#define SKV_SCORE_PD 64                         // row-group loads a wave keeps in flight; >= ITERS: all up front (see the kernel comment)
#endif
// One row group (4 landmark rows of one wave-instruction) against GH query heads held in registers: per-lane fma chain over
// the lane's 8 elements (two heads per v_pk_fma_f32), transposing 16-lane butterfly, one bf16 logit per lane into the tile.
template <int GH>
__device__ __forceinline__ void score_row_group(const f32x2 (&qf)[(GH + 1) / 2][8], const u32x4 xv, bf16_t* sD_g0 /* &sD[g0][0] */,
                                                int col, int lane, float alpha) {
    constexpr int GP = (GH + 1) / 2;
    float xf[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xf[2 * j] = bf_lo(xv[j]);
        xf[2 * j + 1] = bf_hi(xv[j]);
    }
    f32x2 acc2[GP];
#pragma unroll
    for (int gp = 0; gp < GP; ++gp) {
        acc2[gp] = (f32x2){0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc2[gp] = __builtin_elementwise_fma(qf[gp][j], (f32x2){xf[j], xf[j]}, acc2[gp]);
    }
    float part[GH];
#pragma unroll
    for (int g = 0; g < GH; ++g) part[g] = (g & 1) ? acc2[g / 2].y : acc2[g / 2].x;
    const float tot = row16_tree_sum_transposed<GH>(part, lane);
    if (row16_publisher<GH>(lane)) sD_g0[row16_owner<GH>(lane) * SKV_TILE + col] = f2bf(alpha * tot);
}

// ABL (ablation, diagnostic builds only - tools/score_probe.hip): 0 = the kernel; 1 = loads only (no dot
// products); 2 = no per-tile statistics tail.  The library instantiates ABL = 0 only.
//
// WAVES per tile: 16 for G <= 4 (4 / 8 / 16 waves: 9.3 / 8.8 / 8.5 us), 8 for G = 8 (GLM): its q fragment alone is 64 VGPRs,
// 108 in all - with 16 waves only ONE workgroup fits a CU (measured 11.1 vs 9.5 us at the 200K shape); 8 waves fit twice.
// PD = row-group requests a wave keeps in flight (the next one goes out when the oldest has arrived); PD >= ITERS = every
// request up front, the default.  Round-3 probe (tools/score_probe.hip, profiles/r03_score_probe.txt): a bounded depth
// changes nothing (G = 4: 7.75 us up front, 8.2 / 7.7 at depth 2 / 3; G = 8: 9.49 up front, 9.45 at depth 3), i.e. the
// exposed arithmetic is not a late-workgroup effect of first-come-first-served memory service.  The kernel is the
// loads-only time (6.0 us = 5.3 TB/s at G = 4, 5.55 us at G = 8: the HBM rate of a one-wave launch) plus what cannot
// overlap it: the row groups that arrive last (every wave's last request completes near the end of the stream) and the
// per-tile statistics behind the barrier (+1.2 +0.5 us at G = 4; +2.4 +1.5 us at G = 8, where the 144 CUs that hold two
// of the 400 tiles have 3.4 us of VALU work to fit under a 5.5 us stream).
// More than 8 query heads per KV head run as PASSES of 4 over the SAME landmark registers (G = 16 spilled 147 scratch
// operations as one pass): the later passes' query heads wait in a wave-private LDS copy (no barrier).  Results are
// bit-identical: every head's total is the same fma chain and the same 16-lane tree whichever other heads travel through
// the butterfly with it.
template <int G, int ABL = 0, int WAVES = (G == 8 ? 8 : SKV_SCORE_WAVES), int PD = SKV_SCORE_PD>
__global__ __launch_bounds__(64 * WAVES, SKV_SCORE_MIN_WAVES(G, WAVES)) void skv_score_tile_kernel(
    const bf16_t* __restrict__ q,    // [B][G][128]
    const bf16_t* __restrict__ lm,   // [B][N][128]
    bf16_t* __restrict__ D,          // [B][G][N]
    float* __restrict__ part_max,    // [B][T][G]
    float* __restrict__ part_sum,    // [B][T][G]
    int N, int T, float alpha, EarlyHooks eh, FusedSel fs) {
    constexpr int ITERS = 64 / WAVES;               // 4-row wave-instructions per wave
    constexpr int GH = G > 8 ? 4 : G;               // query heads per pass
    constexpr int PASSES = G / GH;
    constexpr int GP = (GH + 1) / 2;
    constexpr int DEPTH = PD < ITERS ? PD : ITERS;
    static_assert(G % GH == 0, "G must be 1, 2, 4, 8 or a multiple of 4");
    const int b = blockIdx.y, t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, rsel = lane >> 4;
    __shared__ __attribute__((aligned(16))) bf16_t sD[G][SKV_TILE];
    // query heads of passes 1.. : one private copy per wave (written and read by the same wave: no barrier)
    __shared__ __attribute__((aligned(16))) bf16_t sQ[PASSES > 1 ? WAVES : 1][PASSES > 1 ? (G - GH) * 128 : 8];

    // q fragment of the first pass: GH x 8 floats, packed in pairs of query heads for v_pk_fma_f32
    f32x2 qf[GP][8];
    auto unpack_q = [&](int gp, const u32x4 w0, const u32x4 w1) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qf[gp][2 * j] = (f32x2){bf_lo(w0[j]), bf_lo(w1[j])};
            qf[gp][2 * j + 1] = (f32x2){bf_hi(w0[j]), bf_hi(w1[j])};
        }
    };
#pragma unroll
    for (int gp = 0; gp < GP; ++gp) {
        u32x4 w0 = *reinterpret_cast<const u32x4*>(q + ((size_t)b * G + 2 * gp) * 128 + 8 * sub);
        u32x4 w1 = (2 * gp + 1 < GH) ? *reinterpret_cast<const u32x4*>(q + ((size_t)b * G + 2 * gp + 1) * 128 + 8 * sub) : w0;
        unpack_q(gp, w0, w1);
    }
    if constexpr (PASSES > 1) {
        constexpr int QV = (G - GH) * 16;           // 16-B units of the later passes' heads
#pragma unroll
        for (int k = 0; k < (QV + 63) / 64; ++k) {
            const int u = lane + 64 * k;
            if (u < QV) reinterpret_cast<u32x4*>(&sQ[wave][0])[u] = reinterpret_cast<const u32x4*>(q + ((size_t)b * G + GH) * 128)[u];
        }
    }

    const bool flag_wave = ABL == 0 && eh.dthr_in != nullptr && wave == (G < WAVES ? G : 0);
    // early fetch: the heads' thresholds.  Uniform addresses (kernel argument + blockIdx): scalar loads into SGPRs, requested
    // here and consumed behind the barrier - no vector register is held across the loop (the G = 4 kernel sits exactly at
    // its 64-VGPR budget) and no wave waits for them
    float dth[G];
#pragma unroll
    for (int g = 0; g < G; ++g) dth[g] = (ABL == 0 && eh.dthr_in != nullptr) ? eh.dthr_in[(size_t)blockIdx.y * G + g] : INFINITY;
    const int row0 = t * SKV_TILE + wave * (4 * ITERS) + rsel;
    u32x4 x[ITERS];
    auto request = [&](int i) __attribute__((always_inline)) {
        int row = row0 + i * 4;
        row = row < N ? row : N - 1;  // clamp: out-of-range rows are computed and discarded
        x[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(lm + ((size_t)b * N + row) * 128 + 8 * sub));
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) request(i);
    if (ABL == 1) {   // memory stream only: fold the loaded words so the loads stay, skip all arithmetic
        uint32_t f = 0;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            if (i + DEPTH < ITERS) request(i + DEPTH);
            f ^= x[i][0] ^ x[i][1] ^ x[i][2] ^ x[i][3];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (f == 0x12345u) D[0] = (bf16_t)f;
        return;
    }
    SCORE_STAMP(0);
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        // program order: request row group i + DEPTH, then consume row group i (the compiler's counted vmcnt wait leaves the
        // DEPTH younger requests in flight); the scheduling barriers keep hipcc from hoisting every request to the top again
        if (i + DEPTH < ITERS) request(i + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
#ifdef SKV_SCORE_STAMPS
        if (i == 0 || i == ITERS - 1) {          // when the first / the last row group's data is there
            if (x[i][0] == 0x12345678u) sD[0][0] = 1;
            SCORE_STAMP(i == 0 ? 1 : 2);
        }
#endif
        score_row_group<GH>(qf, x[i], &sD[0][0], wave * (4 * ITERS) + i * 4 + rsel, lane, alpha);
        __builtin_amdgcn_sched_barrier(0);
    }
    SCORE_STAMP(3);
    if constexpr (PASSES > 1) {
#pragma unroll
        for (int p = 1; p < PASSES; ++p) {
            // the next pass's q fragment must not be built while this one is live, and the stash must really be READ here
            // (forwarding the stored registers would keep all G heads live: that is what the stash is there to avoid)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gp = 0; gp < GP; ++gp) {
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(&sQ[wave][((p - 1) * GH + 2 * gp) * 128 + 8 * sub]);
                const u32x4 w1 = *reinterpret_cast<const u32x4*>(&sQ[wave][((p - 1) * GH + 2 * gp + 1) * 128 + 8 * sub]);
                unpack_q(gp, w0, w1);
            }
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                score_row_group<GH>(qf, x[i], &sD[p * GH][0], wave * (4 * ITERS) + i * 4 + rsel, lane, alpha);
                __builtin_amdgcn_sched_barrier(0);  // one row group at a time (interleaved they need > 64 VGPRs)
            }
        }
    }
    __syncthreads();
    SCORE_STAMP(4);
    if (ABL == 2) {
        if (tid < SKV_TILE && t * SKV_TILE + tid < N) {
#pragma unroll
            for (int g = 0; g < G; ++g) D[((size_t)b * G + g) * N + t * SKV_TILE + tid] = sD[g][tid];
        }
        return;
    }

    // fused selection (fs.ctil != nullptr): per slot the 15-bit key of kappa = max_g (logit_g - ctil_g) and the slot's G logits
    // side by side (Dt [B][N][G]) - every wave takes 256 / WAVES columns, one lane per column (G LDS reads of 2 B, a few VALU
    // operations, one 2-B and one 2G-B store per lane: 64-B / 16G-B runs per wave); D [B][G][N] is not written then
    const bool fused = ABL == 0 && fs.ctil != nullptr;
    if (fused) {
        constexpr int CPW = SKV_TILE / WAVES;                // columns per wave: 16 (G <= 4) or 32 (G = 8)
        const int col = wave * CPW + lane, j = t * SKV_TILE + col;
        if (lane < CPW && j < N) {
            float kap = -INFINITY;
            uint32_t pk[(G + 1) / 2];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bf16_t d = sD[g][col];
                kap = fmaxf(kap, bf2f(d) - fs.ctil[(size_t)b * G + g]);
                if (g & 1) pk[g / 2] |= (uint32_t)d << 16;
                else pk[g / 2] = d;
            }
            // (clamped below zero: K_j = D - c <= 0 always, a kappa above it is a stale ctil - the clamp is monotone, so the
            // candidate rule stands, and it keeps every key of a row within a few hundred codes: one histogram window)
            fs.keys[(size_t)b * fs.key_stride + j] = skv_kappa_key(fminf(kap, -0x1p-12f));
            bf16_t* dt = reinterpret_cast<bf16_t*>(fs.Dt) + ((size_t)b * N + j) * G;
            if constexpr (G == 8) *reinterpret_cast<u32x4*>(dt) = (u32x4){pk[0], pk[1], pk[2], pk[3]};
            else if constexpr (G == 4) *reinterpret_cast<u32x2*>(dt) = (u32x2){pk[0], pk[1]};
            else if constexpr (G == 2) *reinterpret_cast<uint32_t*>(dt) = pk[0];
            else {
#pragma unroll
                for (int g = 0; g < G; ++g) dt[g] = sD[g][col];
            }
        }
    }
    // early-fetch flags (off: eh.dthr_in == nullptr): one wave that has no query head in the statistics below (or wave 0)
    // compares its 4 columns' logits with the heads' thresholds and compacts the flagged slots of the tile
    if (flag_wave) {
        const int c0 = 4 * lane;
        unsigned fl = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float th = dth[g];
            const u32x2 v = *reinterpret_cast<const u32x2*>(&sD[g][c0]);     // the lane's 4 logits of head g: one LDS read
            fl |= (bf_lo(v[0]) >= th ? 1u : 0u) | (bf_hi(v[0]) >= th ? 2u : 0u) | (bf_lo(v[1]) >= th ? 4u : 0u) |
                  (bf_hi(v[1]) >= th ? 8u : 0u);
        }
        if ((t + 1) * SKV_TILE > N) {                        // last tile: columns past the row are not slots
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t * SKV_TILE + c0 + k >= N) fl &= ~(1u << k);
        }
        int base = 0;
        if (__ballot(fl != 0u) != 0ull) {                    // (most tiles of most steps have a few flagged slots; many have none)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long bal = __ballot((fl >> k) & 1u);
                const int pos = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (((fl >> k) & 1u) && pos < SKV_EARLY_K)
                    eh.flag_slot[((size_t)b * T + t) * SKV_EARLY_K + pos] = t * SKV_TILE + c0 + k;
                base += __builtin_popcountll(bal);
            }
        }
        if (lane == 0) eh.flag_cnt[(size_t)b * T + t] = min(base, SKV_EARLY_K);
    }
    // per-tile statistics: wave w owns query head g = w (+ k*waves) and all 256 columns of the tile, 4 consecutive
    // columns per lane, so max, integer exp-sum and the logit store need no cross-wave step and no further barrier
    // (ablation, tools/score_probe.hip: the previous column-per-thread tail cost 3.5 of 11.5 us).
    for (int g = wave; g < G; g += WAVES) {
        const int c0 = 4 * lane;
        float dv[4];
        float mloc = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dv[k] = bf2f(sD[g][c0 + k]);
            if (t * SKV_TILE + c0 + k < N) mloc = fmaxf(mloc, dv[k]);
        }
        const float m = wave_max_dpp(mloc);
        unsigned long long e = 0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t * SKV_TILE + c0 + k < N) e += exp_to_fixed(spec_exp(dv[k] - m));
        e = wave_sum_u64_dpp(e);
        bf16_t* drow = D + ((size_t)b * G + g) * N + (size_t)t * SKV_TILE + c0;
        if (fused) {
            // (the logits went out slot-major above)
        } else if (t * SKV_TILE + c0 + 3 < N && (((size_t)drow) & 7) == 0) {
            *reinterpret_cast<u32x2*>(drow) = *reinterpret_cast<const u32x2*>(&sD[g][c0]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t * SKV_TILE + c0 + k < N) drow[k] = sD[g][c0 + k];
        }
        if (lane == 0) {
            part_max[((size_t)b * T + t) * G + g] = m;
            part_sum[((size_t)b * T + t) * G + g] = fixed_to_float(e);
        }
    }
    SCORE_STAMP(5);
}

// final (max, 1/sum) of one softmax row from its T tile partials; executed by one full wave.
// pm / ps point at element [tile 0] of the row, consecutive tiles are `stride` floats apart.
// The lane-strided accumulation and the reduction tree are part of the arithmetic contract (oracle softmax_finalize);
// the tree's last two levels are taken through readlanes: ((r0 + r1) + (r2 + r3)) of the four row sums, the value every
// lane of the shuffle formulation ends with.
__device__ __forceinline__ void softmax_finalize_wave(const float* pm, const float* ps, int T, int stride,
                                                      int lane, float& m_out, float& inv_out) {
    float m = -INFINITY, acc = 0.0f;
    if (T <= 64) {   // one partial per lane: both loads in flight together (one memory round trip for the finals)
        const bool in = lane < T;
        const float pmv = in ? pm[(size_t)lane * stride] : -INFINITY;
        const float psv = in ? ps[(size_t)lane * stride] : 0.0f;
        m = wave_max_dpp(pmv);
        if (in) acc = acc + psv * spec_exp(pmv - m);
    } else {
        for (int tt = lane; tt < T; tt += 64) m = fmaxf(m, pm[(size_t)tt * stride]);
        m = wave_max_dpp(m);
        for (int tt = lane; tt < T; tt += 64) acc = acc + ps[(size_t)tt * stride] * spec_exp(pm[(size_t)tt * stride] - m);
    }
    acc = row16_tree_sum(acc);
    const int x = __float_as_int(acc);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    const float s = (r0 + r1) + (r2 + r3);
    m_out = m;
    inv_out = 1.0f / s;
}

// legacy stage 2: one wave per (batch, row); writes the finals into the tile-0 slots
// (what the apply stage reads, /root/reference/kernels/batch_gemm_softmax.h:274-275).
__global__ __launch_bounds__(64) void skv_softmax_final_kernel(float* part_max, float* part_sum, int m, int T) {
    const int b = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    float* pm = part_max + (size_t)b * T * m + r;
    float* ps = part_sum + (size_t)b * T * m + r;
    float mx, inv;
    softmax_finalize_wave(pm, ps, T, m, lane, mx, inv);
    // all lanes have finished reading the partials (the reductions above are wave-wide)
    if (lane == 0) {
        pm[0] = mx;
        ps[0] = inv;
    }
}

// legacy stage 3
__global__ __launch_bounds__(256) void skv_softmax_apply_kernel(const bf16_t* __restrict__ D,
                                                                const float* __restrict__ part_max,
                                                                const float* __restrict__ part_sum,
                                                                bf16_t* __restrict__ P, int m, int N, int T) {
    const int b = blockIdx.z, r = blockIdx.y;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= N) return;
    const float mx = part_max[(size_t)b * T * m + r];
    const float inv = part_sum[(size_t)b * T * m + r];
    const size_t o = ((size_t)b * m + r) * N + col;
    P[o] = f2bf(spec_exp(bf2f(D[o]) - mx) * inv);
}

// ---------------------------------------------------------------------------------------
// K2a: finals recomputed per workgroup from the partials (no cross-workgroup hand-off),
// P = bf16(exp(D - m) * inv), score = max over the G query heads of the group.
// ---------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void skv_normalize_groupmax_kernel(
    const bf16_t* __restrict__ D, const float* __restrict__ part_max, const float* __restrict__ part_sum,
    bf16_t* __restrict__ P /* nullable, [B][G][N] */, bf16_t* __restrict__ score /* [B][score_stride] */, int N, int T,
    int score_stride, int tiles_per_block, EarlyHooks eh, int prep_block /* blockIdx.x of the early-fetch role, -1: none */) {
    const int b = blockIdx.y, t0 = blockIdx.x * tiles_per_block;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x == prep_block) {                 // early fetch: flags -> list of chunks to pull (skv_early.h)
        extern __shared__ __attribute__((aligned(16))) int smem_prep[];
        skv_early_prep_role<256>(eh, b, tid, smem_prep);
        return;
    }
    float* const finals_out = eh.finals;
    __shared__ float s_m[G], s_inv[G];
    // the first tile's G logits do not depend on the statistics: request them first (one memory round trip, not two)
    bf16_t dreg[G];
    {
        const int col = t0 * SKV_TILE + tid;
#pragma unroll
        for (int g = 0; g < G; ++g) dreg[g] = col < N ? D[((size_t)b * G + g) * N + col] : (bf16_t)0;
    }
    // every workgroup recomputes the finals from the T partials (no cross-workgroup hand-off); long contexts give a
    // workgroup several tiles so that this stays a small fraction of its work (tiles_per_block, chosen by the launcher)
    constexpr int GW = (G + 3) / 4;     // query heads per wave
    if (T > 64 && T <= 256) {
        // 65..256 tiles (GLM-4 at 200K: 100): the generic routine below makes two dependent passes over the partials per head
        // and a wave owns GW heads one after the other - up to 4 memory round trips on the critical path of a 7 us kernel.
        // Here every partial a wave needs (<= 4 per lane and head) is requested up front, then reduced in the SAME order
        // (maximum first; sum over tiles lane, lane + 64, ... in that order; the contract's reduction tree).
        float pmv[GW][4], psv[GW][4];
#pragma unroll
        for (int k = 0; k < GW; ++k) {
            const int g = wave + 4 * k;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tt = lane + 64 * i;
                const bool in = g < G && tt < T;
                pmv[k][i] = in ? part_max[((size_t)b * T + tt) * G + g] : -INFINITY;
                psv[k][i] = in ? part_sum[((size_t)b * T + tt) * G + g] : 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < GW; ++k) {
            const int g = wave + 4 * k;
            if (g < G) {
                float m = fmaxf(fmaxf(pmv[k][0], pmv[k][1]), fmaxf(pmv[k][2], pmv[k][3]));
                m = wave_max_dpp(m);
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (lane + 64 * i < T) acc = acc + psv[k][i] * spec_exp(pmv[k][i] - m);
                acc = row16_tree_sum(acc);
                const int x = __float_as_int(acc);
                const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
                const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
                if (lane == 0) {
                    s_m[g] = m;
                    s_inv[g] = 1.0f / ((r0 + r1) + (r2 + r3));
                }
            }
        }
    } else {
        for (int g = wave; g < G; g += 4) {
            float mx, inv;
            softmax_finalize_wave(part_max + (size_t)b * T * G + g, part_sum + (size_t)b * T * G + g, T, G, lane, mx, inv);
            if (lane == 0) {
                s_m[g] = mx;
                s_inv[g] = inv;
            }
        }
    }
    __syncthreads();
    if (finals_out != nullptr && blockIdx.x == 0 && tid < G) {
        finals_out[((size_t)b * G + tid) * 2] = s_m[tid];
        finals_out[((size_t)b * G + tid) * 2 + 1] = s_inv[tid];
    }
    for (int tt = 0; tt < tiles_per_block; ++tt) {
        const int col = (t0 + tt) * SKV_TILE + tid;
        if (col >= N) return;
        if (tt > 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) dreg[g] = D[((size_t)b * G + g) * N + col];
        }
        bf16_t best = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t o = ((size_t)b * G + g) * N + col;
            bf16_t p = f2bf(spec_exp(bf2f(dreg[g]) - s_m[g]) * s_inv[g]);
            if (P) P[o] = p;
            best = p > best ? p : best;  // p >= 0: unsigned order == float order
        }
        score[(size_t)b * score_stride + col] = best;
    }
}

// ---------------------------------------------------------------------------------------
// K2b: top-k + diff.  One 1024-thread workgroup per (batch, kv head).
//   1. the bf16 score row (N*2 bytes, 31 KB at the headline config) is staged into LDS with 16-B
//      loads, so the three selection passes never wait on global latency;
//   2. exact k-th value by two 256-bin histograms (high byte, then low byte) - scores are
//      non-negative, so the bf16 pattern orders like the value; the bin search is done by one wave;
//   3. ordered compaction: each thread owns a contiguous index range, one packed block scan gives
//      every element its output position (ties at the threshold go to the lowest index);
//   4. diff against the resident set through an LDS hash set, hits ordered by old slot with a
//      counting pass, misses ordered by chunk id with a rank sort (S^2/1024 compares per thread).
// ---------------------------------------------------------------------------------------
#ifndef SKV_SEL_THREADS
#define SKV_SEL_THREADS 1024
#endif

// inclusive scan of two ints per thread over the 1024-thread workgroup (same barriers for both)
__device__ __forceinline__ void block_scan_incl2(int& a, int& b, int* s_wave /*[32]*/, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    a = wave_scan_incl(a);
    b = wave_scan_incl(b);
    if (lane == 63) {
        s_wave[wave] = a;
        s_wave[16 + wave] = b;
    }
    __syncthreads();
    if (tid < 64) {   // whole wave 0 runs the DPP steps; rows 0 and 1 hold the 16 + 16 wave totals
        int w = tid < 32 ? s_wave[tid] : 0;
        w = row16_scan_incl(w);
        if (tid < 32) s_wave[tid] = w;
    }
    __syncthreads();
    if (wave > 0) {
        a += s_wave[wave - 1];
        b += s_wave[16 + wave - 1];
    }
    __syncthreads();  // s_wave is reused by the next call
}

// Wave 0 only: given a 256-bin histogram (ascending key order) find the bin holding the k-th
// LARGEST element; s_out[0] = bin, s_out[1] = number of elements in higher bins.
__device__ __forceinline__ void select_bin_desc_wave0(const int* hist, int k, int* s_out, int lane) {
    int c[4], t = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = hist[255 - (4 * lane + j)];
        t += c[j];
    }
    const int incl = wave_scan_incl(t);
    int run = incl - t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (run < k && run + c[j] >= k) {
            s_out[0] = 255 - (4 * lane + j);
            s_out[1] = run;
        }
        run += c[j];
    }
}

// SRC: where the selection passes read the score row: 1 = staged in LDS (<= ~60 K scores), 0 = global memory / L2 every
// pass (longer rows: at 131 K scores = 1M-token context the kernel is VALU-bound on its one CU per head, 52 us; holding
// the thread's segment in registers instead was measured 57 us)
template <int SRC>
__global__ __launch_bounds__(SKV_SEL_THREADS) void skv_topk_reorder_kernel(
    const bf16_t* __restrict__ score,      // [B][score_stride] (nullable: then cur_in is used)
    const int64_t* __restrict__ lm_idx,    // [B][N] slot -> chunk id (nullable: identity)
    const int64_t* __restrict__ cur_in,    // [B][S] ids selected by the caller (legacy path)
    int64_t* __restrict__ cached,          // [B][S] in: resident ids per slot; out: reordered ids
    int32_t* __restrict__ offsets,         // [B][S] out
    int32_t* __restrict__ cnts,            // [B] out
    int64_t* __restrict__ sel_out,         // [B][S] nullable: ids selected this step, ascending slot
    int32_t* __restrict__ dst_slots,       // [B][S] nullable.  Non-null = IN-PLACE layout: resident (hit) chunks keep
                                           // their slots, the misses (ascending id) take the freed slots (ascending):
                                           // cached[slot] = id is rewritten only there, offsets[cnt + r] = id and
                                           // dst_slots[cnt + r] = slot of the r-th miss; nothing is written for hits
    int N, int score_stride, int S, int H /* hash size, pow2 >= 2S */, int SP /* pow2 >= S */) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    int* s_cur = smem;            // [SP]
    int* s_hkeys = s_cur + SP;    // [H]
    int* s_hvals = s_hkeys + H;   // [H]
    int* s_byslot = s_hvals + H;  // [SP]  hit key by old slot, -1 if none
    int* s_miss = s_byslot + SP;  // [SP]  misses in selection order
    int* s_rank = s_miss + SP;    // [SP]
    int* s_hist = s_rank + SP;    // [256]
    int* s_wave = s_hist + 256;   // [32]
    int* s_out = s_wave + 32;     // [8]
    int* s_histp = s_out + 8;     // [256][32] lane-privatised pass-1 histogram (copy = lane & 31 -> bank = copy)
    bf16_t* s_score = reinterpret_cast<bf16_t*>(s_histp + 256 * 32);  // [score_stride] when SRC == 1
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the resident id of this thread's slot is requested now: its (cold) latency overlaps the selection passes
    const int my_cached = (tid < S) ? (int)cached[(size_t)b * S + tid] : -1;
    // hash set of the resident ids (key -> lowest slot): initialised here, filled while the scores are in flight
    for (int i = tid; i < H; i += SKV_SEL_THREADS) {
        s_hkeys[i] = -1;
        s_hvals[i] = 0x7fffffff;
    }
    for (int i = tid; i < SP; i += SKV_SEL_THREADS) {
        s_byslot[i] = -1;
        s_rank[i] = 0;
    }
    auto insert_resident = [&]() __attribute__((always_inline)) {
        if (tid < S && my_cached >= 0) {
            unsigned pos = hash_slot(my_cached, H);
            for (int probe = 0; probe < H; ++probe) {
                int prev = atomicCAS(&s_hkeys[pos], -1, my_cached);
                if (prev == -1 || prev == my_cached) {
                    atomicMin(&s_hvals[pos], tid);
                    break;
                }
                pos = (pos + 1) & (unsigned)(H - 1);
            }
        }
    };
    int my_key = -1;   // id selected into position tid (tid < S)

    if (score != nullptr) {
        const bf16_t* gsc = score + (size_t)b * score_stride;
        const u32x4* gvec = reinterpret_cast<const u32x4*>(gsc);
        const u32x4* svec = SRC == 1 ? reinterpret_cast<const u32x4*>(s_score) : gvec;
        const int nvec = score_stride / 8;
        // thread-owned contiguous segment of `segv` 8-score vectors (ordered compaction)
        const int segv = (nvec + SKV_SEL_THREADS - 1) / SKV_SEL_THREADS;
        const int v0 = min(tid * segv, nvec), v1 = min(v0 + segv, nvec);
        TOPK_STAMP(0);
        for (int i = tid; i < 256 * 32; i += SKV_SEL_THREADS) s_histp[i] = 0;
        __syncthreads();
        // ---- pass 1: histogram of the high byte, fused with the staging copy.  Softmax probabilities of one head
        // span a handful of binades, so nearly all 15 K scores fall into 2-4 high-byte bins: atomics on ONE 256-bin
        // histogram serialise 64-way per wave instruction (measured 7.2 us for this pass on log-normal scores).  The
        // histogram is therefore privatised 32 ways by lane: copy c = lane & 31 of bin b lives at [b*32 + c], i.e. in
        // bank c, so a wave instruction is conflict-free whatever the value distribution (lanes l and l+32 pair up).
        // A vector whose 8 scores share their high byte adds 8 with one atomic.
        const int cpy = lane & 31;
        auto hist1 = [=](int i, const u32x4 v) __attribute__((always_inline)) {
            const int b0 = (int)((v[0] >> 8) & 0xff);
            bool uniform = i * 8 + 7 < N;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                uniform = uniform && (int)((v[j] >> 8) & 0xff) == b0 && (int)(v[j] >> 24) == b0;
            if (uniform) {
                atomicAdd(&s_histp[b0 * 32 + cpy], 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int val = (e & 1) ? (int)(v[e >> 1] >> 16) : (int)(v[e >> 1] & 0xffffu);
                    if (i * 8 + e < N) atomicAdd(&s_histp[(val >> 8) * 32 + cpy], 1);
                }
            }
        };
        for (int i = tid; i < nvec; i += SKV_SEL_THREADS) {
            const u32x4 v = gvec[i];
            if (SRC == 1) reinterpret_cast<u32x4*>(s_score)[i] = v;
            hist1(i, v);
        }
        insert_resident();   // (hash arrays were initialised before the barrier above)
        __syncthreads();
        if (tid < 256) {   // fold the 32 copies; thread t starts at copy t so the 64 lanes of a wave read 32 banks
            int t = 0;
#pragma unroll
            for (int r = 0; r < 32; ++r) t += s_histp[tid * 32 + ((r + tid) & 31)];
            s_hist[tid] = t;
        }
        __syncthreads();
        TOPK_STAMP(1);
        if (wave == 0) select_bin_desc_wave0(s_hist, S, s_out, lane);
        __syncthreads();
        const int hi = s_out[0], above_hi = s_out[1];
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        TOPK_STAMP(2);
        // ---- pass 2: histogram of the low byte inside that bin (values spread over many bins: plain atomics)
        auto hist2 = [=](int i, const u32x4 v) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int val = (e & 1) ? (int)(v[e >> 1] >> 16) : (int)(v[e >> 1] & 0xffffu);
                if ((i * 8 + e < N) && ((val >> 8) == hi)) atomicAdd(&s_hist[val & 0xff], 1);
            }
        };
        for (int i = tid; i < nvec; i += SKV_SEL_THREADS) hist2(i, svec[i]);
        __syncthreads();
        TOPK_STAMP(3);
        if (wave == 0) select_bin_desc_wave0(s_hist, S - above_hi, s_out + 2, lane);
        __syncthreads();
        const int thr = (hi << 8) | s_out[2];
        const int need_eq = S - (above_hi + s_out[3]);
        TOPK_STAMP(4);
        // ---- pass 3: ordered compaction over the thread's segment
        // (the lambdas below take and return their counters by value: captured references would pin them to scratch)
        struct Cnt { int gt, eq; };
        auto count = [=](int i, const u32x4 v, Cnt c) __attribute__((always_inline)) -> Cnt {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int val = (e & 1) ? (int)(v[e >> 1] >> 16) : (int)(v[e >> 1] & 0xffffu);
                const bool in = i * 8 + e < N;
                c.gt += in && val > thr;
                c.eq += in && val == thr;
            }
            return c;
        };
        Cnt cc{0, 0};
        for (int i = v0; i < v1; ++i) cc = count(i, svec[i], cc);
        const int c_gt = cc.gt, c_eq = cc.eq;
        TOPK_STAMP(5);
        int gt_before = c_gt, eq_before = c_eq;
        block_scan_incl2(gt_before, eq_before, s_wave, tid);
        TOPK_STAMP(6);
        gt_before -= c_gt;
        eq_before -= c_eq;
        // one candidate: exact rank among the selected (ties at the threshold -> lowest slot first)
        auto assign = [=](int i, const u32x4 v, Cnt c) __attribute__((always_inline)) -> Cnt {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int val = (e & 1) ? (int)(v[e >> 1] >> 16) : (int)(v[e >> 1] & 0xffffu);
                const int j = i * 8 + e;
                if (j >= N) continue;
                int pos = -1;
                if (val > thr) {
                    pos = c.gt + min(c.eq, need_eq);
                    ++c.gt;
                } else if (val == thr) {
                    if (c.eq < need_eq) pos = c.gt + c.eq;
                    ++c.eq;
                }
                if (pos >= 0) s_cur[pos] = j;
            }
            return c;
        };
        Cnt before{gt_before, eq_before};
        for (int i = v0; i < v1; ++i) before = assign(i, svec[i], before);
        __syncthreads();
        // slot -> chunk id for the S selected slots: one parallel gather (not inside the serial loop above)
        if (tid < S) {
            const int j = s_cur[tid];
            const long long id = lm_idx ? lm_idx[(size_t)b * N + j] : (long long)j;
            my_key = (int)id;
            if (sel_out) sel_out[(size_t)b * S + tid] = id;
        }
    } else {
        if (tid < S) my_key = (int)cur_in[(size_t)b * S + tid];
        __syncthreads();     // hash arrays initialised
        insert_resident();
        __syncthreads();
    }

    TOPK_STAMP(7);
    TOPK_STAMP(8);
    // ---- classify the new ids
    int my_slot = -1;
    if (tid < S) {
        if (my_key >= 0) {
            unsigned pos = hash_slot(my_key, H);
            for (int probe = 0; probe < H; ++probe) {
                int k2 = s_hkeys[pos];
                if (k2 == my_key) {
                    my_slot = s_hvals[pos];
                    break;
                }
                if (k2 == -1) break;
                pos = (pos + 1) & (unsigned)(H - 1);
            }
        }
        if (my_slot >= 0) s_byslot[my_slot] = my_key;
    }
    __syncthreads();
    TOPK_STAMP(9);
    // hits ordered by old slot (compaction of s_byslot), misses in selection order
    const int is_hit_slot = (tid < S && s_byslot[tid] >= 0) ? 1 : 0;
    const int is_miss = (tid < S && my_slot < 0) ? 1 : 0;
    int hit_incl = is_hit_slot, miss_incl = is_miss;
    block_scan_incl2(hit_incl, miss_incl, s_wave, tid);
    if (tid == SKV_SEL_THREADS - 1) {
        s_out[4] = hit_incl;
        s_out[5] = miss_incl;
    }
    if (is_miss) s_miss[miss_incl - 1] = my_key;
    int* s_free = s_hvals;  // the hash values are dead after the classification: r-th free slot (ascending)
    if (dst_slots && tid < S && !is_hit_slot) s_free[tid - hit_incl] = tid;   // free slots among 0..tid: tid+1-hit_incl
    __syncthreads();
    const int cnt = s_out[4], nm = s_out[5];
    TOPK_STAMP(10);
    // misses ordered by chunk id.  They are collected in selection order (ascending landmark slot) and the
    // slot -> chunk-id map is normally increasing, so they are usually sorted already: one vote decides; the
    // rank sort (ties by position, P threads share one element) only runs otherwise.
    const int unsorted_here = (tid > 0 && tid < nm && s_miss[tid - 1] > s_miss[tid]) ? 1 : 0;
    if (__syncthreads_or(unsorted_here)) {
        const int P = SKV_SEL_THREADS / SP;
        const int i = tid / P, part = tid % P;
        if (i < nm) {
            const int ki = s_miss[i];
            int r = 0;
            for (int j = part; j < nm; j += P) {
                int kj = s_miss[j];
                r += (kj < ki) || (kj == ki && j < i);
            }
            if (r) atomicAdd(&s_rank[i], r);
        }
    } else if (tid < nm) {
        s_rank[tid] = tid;
    }
    __syncthreads();
    TOPK_STAMP(11);
    // ---- write out
    if (dst_slots) {
        if (tid < nm) {
            const int key = s_miss[tid], r = s_rank[tid], slot = s_free[r];
            cached[(size_t)b * S + slot] = (long long)key;
            offsets[(size_t)b * S + cnt + r] = key;
            dst_slots[(size_t)b * S + cnt + r] = slot;
        }
        if (tid == 0) cnts[b] = cnt;
        return;
    }
    if (is_hit_slot) {
        int o = hit_incl - 1;
        cached[(size_t)b * S + o] = (long long)s_byslot[tid];
        offsets[(size_t)b * S + o] = tid;
    }
    if (tid < nm) {
        int key = s_miss[tid], o = cnt + s_rank[tid];
        cached[(size_t)b * S + o] = (long long)key;
        offsets[(size_t)b * S + o] = key;
    }
    if (tid == 0) cnts[b] = cnt;
    TOPK_STAMP(12);
}

// ---------------------------------------------------------------------------------------
// Fused selection front end (round 4): exact top-S WITHOUT the normalise launch.
//
// The three-launch path needs score_j = max_g bf16(e(D_gj - m_g) inv_g) for every slot, i.e. the finals (m_g, 1 / s_g) of the
// softmax rows BEFORE any score exists - a grid-wide dependency that costs a launch (6.1 us of a 57 us chain, during which the
// PCIe link, the roof of the path, sits idle).  But which slots can be in the top S is decided by far fewer of them:
//   * K_j = max_g (D_gj - c_g), c_g = m_g + ln s_g, is score_j in the logit domain: score_j ~ exp(K_j) up to the bf16
//     rounding of P and the f32 rounding of e() and 1 / s (relative 2^-9 + ~1e-6);
//   * the scan launch cannot know c_g, but it knows the PREVIOUS step's: it emits kappa_j = max_g (D_gj - ctil_g) as a 15-bit
//     monotone key (skv_kappa_key).  With delta_g = c_g - ctil_g:  kappa_j - max delta <= K_j <= kappa_j - min delta;
//   * let k15 = the S-th largest key: at least S slots a have kappa_a >= low(k15) (skv_kappa_key_low).  A slot j with
//       kappa_j < theta := low(k15) - (max delta - min delta) - SLACK
//     has K_j <= kappa_j - min delta < low(k15) - max delta - SLACK <= K_a - SLACK for those S slots a, hence
//       score_j <= bf16(exp(K_j)(1 + 2e-6)) <= exp(K_a) e^-SLACK (1 + 2^-9 + 2e-6) < exp(K_a)(1 - 2^-9 - 2e-6) <= score_a
//     for SLACK = 2^-4 (e^-0.0625 = 0.939) whenever score_a is a normal bf16 number: j is STRICTLY below S slots - it is
//     neither in the top S nor tied with its last member, whatever the tie rule.  key(kappa_j) < key(theta) implies
//     kappa_j < theta (monotone key), so the candidates are the slots whose key reaches key(theta);
//   * the candidates (S plus a few dozen when the c_g move together, as they do from step to step; all that matters is the
//     SPREAD of the delta_g, a common shift cancels) are evaluated EXACTLY - the finals by the contract's reduction
//     (softmax_finalize), P = bf16(spec_exp(D - m_g) inv_g), max over g: the normalise kernel's arithmetic - and the exact
//     top S with ties to the lowest slot is taken among them (second threshold search, ordered placement).
// More than T3_CAND candidates (first step: ctil = 0; a jump of the query), or an S-th score that is not a normal bf16
// number: every thread evaluates ALL its slots exactly and the standard front end runs on those keys - same result, the
// three-launch path's arithmetic, ~5 us slower on that step.  Nothing is approximate: oracle/shadowkv_oracle.c
// (oracle_fused_candidates) restates the candidate rule and the tests check top-S subset-of candidates on every input.
// Returns true when s_cur / s_id hold the selection; false: w[] holds the exact score keys (padding 0), histogram zeroed.
// ---------------------------------------------------------------------------------------
template <int FG>
__device__ __forceinline__ uint32_t t3_exact_key(const bf16_t* __restrict__ drow, const float* s_fin) {
    bf16_t d[FG];
    if constexpr (FG == 8) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(drow);
#pragma unroll
        for (int x = 0; x < 4; ++x) { d[2 * x] = (bf16_t)(v[x] & 0xffffu); d[2 * x + 1] = (bf16_t)(v[x] >> 16); }
    } else if constexpr (FG == 4) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(drow);
#pragma unroll
        for (int x = 0; x < 2; ++x) { d[2 * x] = (bf16_t)(v[x] & 0xffffu); d[2 * x + 1] = (bf16_t)(v[x] >> 16); }
    } else {
#pragma unroll
        for (int g = 0; g < FG; ++g) d[g] = drow[g];
    }
    bf16_t best = 0;
#pragma unroll
    for (int g = 0; g < FG; ++g) {       // the normalise kernel's arithmetic (skv_normalize_groupmax_kernel), bit for bit
        const bf16_t p = f2bf(spec_exp(bf2f(d[g]) - s_fin[2 * g]) * s_fin[2 * g + 1]);
        best = p > best ? p : best;      // p >= 0: unsigned order == float order
    }
    return best;
}

template <int FG, int NW, typename F>
__device__ __forceinline__ bool t3_fused_front(uint32_t (&w)[NW], const FusedTop& ft, const int64_t* __restrict__ lm_idx, const int b,
                                               const int N, const int S, const int j0, const int tid, int* s_hist, int* s_w,
                                               int* s_out, int* s_cur, long long* s_id, int* s_cand, float* s_fin,
                                               const EarlyHooks& eh, F insert_resident) {
    constexpr int NG = (NW + 15) / 16;
    const int lane = tid & 63, wave = tid >> 6, T = ft.T;
    const bf16_t* const Dt = reinterpret_cast<const bf16_t*>(ft.Dt) + (size_t)b * N * FG;
    // (1) the tile partials of softmax row g = wave (<= 4 per lane: T <= 256), reduced in the contract's order: maximum; sum over
    //     tiles lane, lane + 64, .. in that order; the 16-lane tree; the four rows
    float pmv[4], psv[4];
    float my_ctil = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tt = lane + 64 * i;
        const bool in = wave < FG && tt < T;
        pmv[i] = in ? ft.part_max[((size_t)b * T + tt) * FG + wave] : -INFINITY;
        psv[i] = in ? ft.part_sum[((size_t)b * T + tt) * FG + wave] : 0.0f;
    }
    if (wave < FG) my_ctil = ft.ctil[(size_t)b * FG + wave];
    const int kguess = ft.level[b];                              // witness level carried over from the previous step (0: none)
    if (wave < FG) {
        float m = fmaxf(fmaxf(pmv[0], pmv[1]), fmaxf(pmv[2], pmv[3]));
        m = wave_max_dpp(m);
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < T) acc = acc + psv[i] * spec_exp(pmv[i] - m);
        acc = row16_tree_sum(acc);
        const int x = __float_as_int(acc);
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
        if (lane == 0) {
            const float inv = 1.0f / ((r0 + r1) + (r2 + r3));
            const float c = m - __logf(inv);                     // c_g = m_g + ln s_g (prediction + margin only)
            s_fin[2 * wave] = m;
            s_fin[2 * wave + 1] = inv;
            s_fin[2 * FG + wave] = c - my_ctil;                  // delta_g
            ft.ctil[(size_t)b * FG + wave] = c;                  // the next step's scan takes its keys against this
        }
    }
    __syncthreads();                                             // finals visible; the caller's LDS initialisation complete
    insert_resident();                                           // (hash set of the resident ids: LDS atomics beside the mask work)
    TOPK_STAMP(12);
    float spread;
    {
        float dmax = -INFINITY, dmin = INFINITY;
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            dmax = fmaxf(dmax, s_fin[2 * FG + g]);
            dmin = fminf(dmin, s_fin[2 * FG + g]);
        }
        spread = dmax - dmin;
    }
    // candidates for a witness level `lvl` (at least S keys reach it - to be verified): every key >= key(low(lvl) - spread - 2^-4).
    // One packed block scan counts the keys >= lvl (low half) and the candidates (high half) and places the candidates'
    // slots, in slot order, in s_cand.  Returns the two totals packed the same way.
    uint32_t mc[NG];
    auto candidates = [&](const int lvl, int* scan_row) __attribute__((always_inline)) -> int {
        const float theta = skv_kappa_key_low(lvl) - spread - 0.0625f;
        int thr_lo = (theta == theta) ? (int)skv_kappa_key(theta) : 0;       // (NaN: every slot is a candidate)
        thr_lo = min(thr_lo, lvl);
        const uint32_t kge = (0x8000u - (uint32_t)thr_lo) * 0x10001u;       // keys are < 0x8000 (15 bits)
        const uint32_t kgl = (0x8000u - (uint32_t)lvl) * 0x10001u;
        uint32_t ml[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            mc[g] = 0u;
            ml[g] = 0u;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const uint32_t te = w[i] + kge, tl = w[i] + kgl;
            mc[i / 16] |= (((te >> 15) & 1u) | ((te >> 31) << 1)) << (2 * (i % 16));
            ml[i / 16] |= (((tl >> 15) & 1u) | ((tl >> 31) << 1)) << (2 * (i % 16));
        }
        if (j0 + NW * 2 > N) {                                   // padding (key 0) is never a candidate, whatever the level
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int j = j0 + 2 * i;
                const uint32_t dead = ((j >= N ? 1u : 0u) | (j + 1 >= N ? 2u : 0u)) << (2 * (i % 16));
                mc[i / 16] &= ~dead;
                ml[i / 16] &= ~dead;
            }
        }
        int cc = 0, cl = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cc += __builtin_popcount(mc[g]);
            cl += __builtin_popcount(ml[g]);
        }
        const int pk = cl | (cc << 16);                          // (each total <= 32,768: per thread <= 32 keys)
        const int pincl = block_scan_incl1(pk, scan_row, tid);
        if (tid == T2_THREADS - 1) s_out[8] = pincl;
        const int cincl = (int)((unsigned)pincl >> 16);
        if (cincl <= T3_CAND) {                                  // (a thread beyond the limit implies a total beyond it)
            int o = cincl - cc;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint32_t m = mc[g];
                while (m) {
                    s_cand[o++] = j0 + g * 32 + __builtin_ctz(m);
                    m &= m - 1;
                }
            }
        }
        __syncthreads();
        return s_out[8];
    };
    // (2) the witness level.  Common case: the level the previous step left behind still has at least S keys above it (the
    // k-th largest score hardly moves from step to step) and few enough candidates - no search at all.  Otherwise (first step,
    // a jump of the query): the exact S-th largest key, by the histogram search.
    int level = 0, C = T3_CAND + 1, n_lvl = 0;
    bool searched = false;
    if (kguess >= 8 && kguess < 0x8000) {
        const int tot = candidates(kguess, s_w + 32);
        n_lvl = tot & 0xffff;
        C = (int)((unsigned)tot >> 16);
        level = kguess;
    }
    TOPK_STAMP(13);
    if (n_lvl < S || C > T3_CAND) {
        int thr15, ne15;
        t2_find_threshold<NW>(w, T2_THREADS * (NW / 4) * 8 - N, S, tid, s_hist, s_w, s_out, thr15, ne15, [] {});
        searched = true;
        {   // the histogram is searched again below: all its readers are behind find_threshold's last barrier
            u32x4* hz = reinterpret_cast<u32x4*>(s_hist);
#pragma unroll
            for (int k = 0; k < T2_BINS * T2_COPIES / 4 / T2_THREADS; ++k) hz[tid + k * T2_THREADS] = (u32x4){0u, 0u, 0u, 0u};
        }
        const int tot = candidates(thr15, s_w + 48);
        n_lvl = tot & 0xffff;
        C = (int)((unsigned)tot >> 16);
        level = thr15;
    }
    TOPK_STAMP(14);
    if (tid == 0) {
        // the next step's level (codes of 1/256 in kappa): 0.09 below the S-th key when it was searched; otherwise steered so that
        // S + S/16 .. S + S/2 keys stay above it - a level that is too high fails the witness count and costs that step a search,
        // one that is too low only costs candidates (free up to T3_CAND: two per thread either way)
        int nl = searched ? level - 24 : (n_lvl > S + S / 2 ? level + 6 : n_lvl < S + S / 16 ? level - 12 : level);
        ft.level[b] = min(max(nl, 8), 32767);
        ft.stats[2 * b] = (searched ? 1 : 0) | (C > T3_CAND ? 2 : 0);     // (diagnostics: skv_select_state_stats_offset)
        ft.stats[2 * b + 1] = C;
    }
    TOPK_STAMP(15);
    if (C <= T3_CAND) {
        // (4) exact scores of the candidates, two per thread in slot order; their chunk ids travel with them
        const int c0 = 2 * tid, c1 = 2 * tid + 1;
        const int ja = c0 < C ? s_cand[c0] : -1, jb = c1 < C ? s_cand[c1] : -1;
        long long ida = -1, idb = -1;
        if (lm_idx != nullptr) {
            ida = lm_idx[(size_t)b * N + max(ja, 0)];
            idb = lm_idx[(size_t)b * N + max(jb, 0)];
        } else {
            ida = ja;
            idb = jb;
        }
        const uint32_t ka = t3_exact_key<FG>(Dt + (size_t)max(ja, 0) * FG, s_fin);
        const uint32_t kb = t3_exact_key<FG>(Dt + (size_t)max(jb, 0) * FG, s_fin);
        const uint32_t w2[1] = {(ja >= 0 ? ka : 0u) | ((jb >= 0 ? kb : 0u) << 16)};
#ifdef SKV_TOPK_STAMPS
        if (w2[0] == 0x12345678u) s_out[10] = 1;     // waits for the gathered logits
        TOPK_STAMP(16);
#endif
        int thr, need_eq;
        // (with the early state: also the (S + SKV_NEAR_MAX)-th candidate score - the near misses below are the candidates between the two)
        int thr_near[2] = {0, 0};
        t2_find_threshold<1>(w2, 2 * T2_THREADS - C, S, tid, s_hist, s_w, s_out, thr, need_eq, [] {},
                             eh.near_ids != nullptr ? S + SKV_NEAR_MAX : 0, thr_near);
        TOPK_STAMP(17);
        if (thr >= 0x0100) {                                     // a normal bf16 number: the strictness argument holds
            if (eh.dthr_out != nullptr && tid < FG) {            // next step's flag thresholds (early fetch; prediction only)
                const float kth = __uint_as_float((uint32_t)thr << 16);
                const float mx = s_fin[2 * tid], inv = s_fin[2 * tid + 1];
                eh.dthr_out[(size_t)b * FG + tid] = (kth > 0.f && inv > 0.f) ? mx + __logf(kth / inv) + eh.margin : INFINITY;
            }
            const int lo = (int)(w2[0] & 0xffffu), hi = (int)(w2[0] >> 16);
            const bool va = ja >= 0, vb = jb >= 0;
            const int ga = va && lo > thr, gb = vb && hi > thr, ea = va && lo == thr, eb = vb && hi == thr;
            const int packed = (ga + gb) | ((ea + eb) << 12);    // (<= 1,024 greater, <= 2,048 equal in all)
            const int pincl = block_scan_incl1(packed, searched ? s_w + 32 : s_w + 48, tid);
            const int pexcl = pincl - packed;
            int gt_run = pexcl & 0xfff, eq_run = pexcl >> 12;
            if (ga) {
                const int pos = gt_run + min(eq_run, need_eq);
                s_cur[pos] = ja;
                s_id[pos] = ida;
                ++gt_run;
            } else if (ea) {
                if (eq_run < need_eq) {
                    s_cur[gt_run + eq_run] = ja;
                    s_id[gt_run + eq_run] = ida;
                }
                ++eq_run;
            }
            if (gb) {
                const int pos = gt_run + min(eq_run, need_eq);
                s_cur[pos] = jb;
                s_id[pos] = idb;
            } else if (eb && eq_run < need_eq) {
                s_cur[gt_run + eq_run] = jb;
                s_id[gt_run + eq_run] = idb;
            }
            TOPK_STAMP(18);
            if (eh.near_ids != nullptr) {
                // Near misses (round 5; prediction only, nothing downstream of the selection reads it): the candidates that were
                // evaluated exactly and fell short of the S-th score are the chunks most likely to enter the NEXT step's selection
                // (tools/near_miss_sim.py: a third of the 64 nearest do, a quarter of the next 64).  The candidates from the S-th
                // score down to the (S + 64)-th (thr_near[0]) go to list 0, those down to the (S + 128)-th to list 1 (0 = every
                // candidate, when that score is outside the histogram window), the first SKV_NEAR_MAX in slot order each; the
                // gate/up GEMV launch of this layer stages list 0 while the link is idle, the down-projection launch list 1
                // (skv_near_pull_role).  One more block scan: the selection workgroup finishes microseconds before the launch's
                // pull workgroups.
                // list 0: scores in [thr_near[0], thr) - ranks S + 1 .. S + 64; list 1: [thr_near[1], thr_near[0]) - ranks S + 65 .. S + 128
                const int la = va && lo < thr, lb = vb && hi < thr;
                const int a0 = la && lo >= thr_near[0], b0 = lb && hi >= thr_near[0];
                const int a1 = la && !a0 && lo >= thr_near[1], b1 = lb && !b0 && hi >= thr_near[1];
                const int npk = (a0 + b0) | ((a1 + b1) << 16);
                const int nincl = block_scan_incl1(npk, searched ? s_w + 48 : s_w + 32, tid);   // (the row the last scan did not use)
                int p0 = (nincl & 0xffff) - (a0 + b0), p1 = (nincl >> 16) - (a1 + b1);
                int* const l0 = eh.near_ids + (size_t)b * SKV_NEAR_MAX;
                int* const l1 = eh.near_ids + ((size_t)eh.near_B + b) * SKV_NEAR_MAX;
                if (a0) { if (p0 < SKV_NEAR_MAX) l0[p0] = (int)ida; ++p0; }
                if (b0 && p0 < SKV_NEAR_MAX) l0[p0] = (int)idb;
                if (a1) { if (p1 < SKV_NEAR_MAX) l1[p1] = (int)ida; ++p1; }
                if (b1 && p1 < SKV_NEAR_MAX) l1[p1] = (int)idb;
                if (tid == T2_THREADS - 1) {
                    eh.near_cnt[b] = min(nincl & 0xffff, SKV_NEAR_MAX);
                    eh.near_cnt[eh.near_B + b] = min(nincl >> 16, SKV_NEAR_MAX);
                }
            }
            return true;
        }
        if (tid == 0) ft.stats[2 * b] |= 2;
        u32x4* hz = reinterpret_cast<u32x4*>(s_hist);            // (never for softmax scores of a real head)
#pragma unroll
        for (int k = 0; k < T2_BINS * T2_COPIES / 4 / T2_THREADS; ++k) hz[tid + k * T2_THREADS] = (u32x4){0u, 0u, 0u, 0u};
    }
    // (5) slow path: the exact score of EVERY slot of the thread; the caller runs the standard front end on these keys
#pragma unroll 1
    for (int i = 0; i < NW; ++i) {
        const int j = j0 + 2 * i;
        const uint32_t k0 = j < N ? t3_exact_key<FG>(Dt + (size_t)j * FG, s_fin) : 0u;
        const uint32_t k1 = j + 1 < N ? t3_exact_key<FG>(Dt + (size_t)(j + 1) * FG, s_fin) : 0u;
        w[i] = k0 | (k1 << 16);
    }
    return false;
}

// ---------------------------------------------------------------------------------------
// K2b, second generation (default for N <= 131,072 scores per head): same results as the kernel above with a
// shorter dependency chain and far fewer instructions per score.  One 1,024-thread workgroup per head runs on ONE
// CU: every instruction a thread executes is issued 16 times on 4 SIMDs (about 20 cycles), so the kernel is bound by
// instructions per thread (in-kernel stamps, tools/topk_probe.hip: histogram 1.45 us, count 1.2 us, ordered
// compaction 1.95 us of 9.8 us before this rewrite), not by the 31 KB it reads.
//   * the thread's scores stay in registers (SEGV contiguous 16-B vectors per thread, loaded once), two 16-bit keys
//     per word, and are processed two at a time with packed 16-bit arithmetic;
//   * ONE histogram pass finds the exact k-th value: 4,096 bins of the 16-bit pattern counted down from the head's
//     maximum (32 binades; softmax scores of one head fit - otherwise the window slides and the pass repeats), 4
//     lane-private copies per bin laid out [bin][copy].  Real rows spread over hundreds of these bins, so the
//     serialisation that forced a 32-way privatised 256-bin histogram + a second pass above does not arise;
//   * "greater" / "greater or equal" flags of the thread's scores are bit masks (one add + three bit operations per
//     word and mask); counts are popcounts, and the ordered compaction loops over the SET bits only (256 of 15 K
//     scores are selected);
//   * the slot -> chunk-id loads of a thread's first two candidates are issued as soon as the k-th value is known
//     and travel with the candidate, so the (HBM-cold) gather overlaps the block scan;
//   * block scans keep one barrier (every wave re-scans the 16 wave totals itself), both counters of a scan are
//     packed into one integer.
// Padding (scores past N in the last thread's vectors) is rewritten to key 0 at load and its count is subtracted
// from the bin of key 0; padding has the highest indices, so the tie rule never reaches it.
// ---------------------------------------------------------------------------------------
template <int SEGV, int FG = 0 /* query heads per KV head of the fused front end; 0: scores in */>
__global__ __launch_bounds__(T2_THREADS) void skv_topk2_kernel(
    const bf16_t* __restrict__ score,      // [B][score_stride] (nullable: then cur_in is used)
    const int64_t* __restrict__ lm_idx,    // [B][N] slot -> chunk id (nullable: identity)
    const int64_t* __restrict__ cur_in,    // [B][S] ids selected by the caller (legacy path)
    int64_t* __restrict__ cached,          // [B][S] in: resident ids per slot; out: reordered ids
    int32_t* __restrict__ offsets, int32_t* __restrict__ cnts, int64_t* __restrict__ sel_out,
    int32_t* __restrict__ dst_slots,       // nullable; non-null = in-place layout (see skv_topk_reorder_kernel)
    int N, int score_stride, int S, int H /* hash size, pow2 >= 4R */, int SP /* pow2 >= S */,
    // resident set larger than the selection (in-place layout only): R slots per head (S <= R <= 1024), `cached` is
    // [B][R], the S - cnt misses replace the least recently selected of the R - cnt slots that were not selected now
    // (age = steps since the slot's chunk was last selected, saturating at 62; 63 = empty slot; ties -> lowest slot).
    // R == S: every slot that was not selected is replaced, slot_age is not touched (may be null).
    int R, int RP /* pow2 >= R */, int32_t* __restrict__ slot_age /* [B][R] */, EarlyHooks eh, FusedTop ft) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    int* s_hist = smem;                               // [T2_BINS][T2_COPIES]
    int* s_cur = s_hist + T2_BINS * T2_COPIES;        // [SP]   selected landmark slot per output position
    long long* s_id = reinterpret_cast<long long*>(s_cur + SP);   // [SP] its chunk id (-1: not gathered yet)
    int* s_hkeys = reinterpret_cast<int*>(s_id + SP);  // [H]
    int* s_hvals = s_hkeys + H;                       // [H]
    int* s_byslot = s_hvals + H;                      // [RP]
    int* s_age = s_hist;                              // [64] age histogram (the score histogram is dead by then)
    int* s_miss = s_byslot + RP;                      // [SP]
    int* s_rank = s_miss + SP;                        // [SP]
    int* s_w = s_rank + SP;                           // [4][16] wave totals (one row per block scan) + [16] wave maxima
    int* s_out = s_w + 80;                            // [16]
    int* s_cand = s_out + 16;                         // [T3_CAND] candidate slots (fused front end)
    float* s_fin = reinterpret_cast<float*>(s_cand + (FG > 0 ? T3_CAND : 0));   // [3 * FG] finals (m, 1 / s) and deltas
    const int tid = threadIdx.x;
    if (eh.staging != nullptr && (int)blockIdx.x >= (int)gridDim.x / (1 + eh.pull_wgs)) {
        // early fetch: the blocks behind the B selection blocks pull (skv_early.h)
        const int p = (int)blockIdx.x - (int)gridDim.x / (1 + eh.pull_wgs);
        if constexpr (FG > 0) skv_early_prep_pull_role<T2_THREADS>(eh, p / eh.pull_wgs, p % eh.pull_wgs, tid, smem);   // no normalise launch: list + pull
        else skv_early_pull_role<T2_THREADS>(eh, p / eh.pull_wgs, p % eh.pull_wgs, tid, smem);
        return;
    }
    const int b = blockIdx.x;
    TOPK_STAMP(0);
    constexpr int NW = 4 * SEGV;                      // 32-bit words (two keys each) per thread
    constexpr int NG = (NW + 15) / 16;                // mask registers: 16 words (32 keys) each
    // everything whose address is known is requested first: the resident id of this thread's slot, the thread's scores
    const int my_cached = (tid < R) ? (int)cached[(size_t)b * R + tid] : -1;
    const int my_age = (R > S && tid < R) ? slot_age[(size_t)b * R + tid] : 0;    // consumed after the classification
    uint32_t w[NW];
    const int j0 = tid * SEGV * 8;                    // first score index of this thread
    if (score != nullptr) {
        const u32x4* gvec = reinterpret_cast<const u32x4*>(score + (size_t)b * score_stride);
        const int nvec = score_stride / 8;
#pragma unroll
        for (int k = 0; k < SEGV; ++k) {
            const int vi = tid * SEGV + k;
            const u32x4 v = vi < nvec ? gvec[vi] : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int x = 0; x < 4; ++x) w[4 * k + x] = v[x];
        }
    }
    {   // LDS init: histogram (16 words per thread), hash set, slot arrays
        u32x4* hz = reinterpret_cast<u32x4*>(s_hist);
#pragma unroll
        for (int k = 0; k < T2_BINS * T2_COPIES / 4 / T2_THREADS; ++k) hz[tid + k * T2_THREADS] = (u32x4){0u, 0u, 0u, 0u};
        for (int i = tid; i < H; i += T2_THREADS) {
            s_hkeys[i] = -1;
            s_hvals[i] = 0x7fffffff;
        }
        for (int i = tid; i < RP; i += T2_THREADS) s_byslot[i] = -1;
        for (int i = tid; i < SP; i += T2_THREADS) s_rank[i] = 0;
        if (tid == 0) {
            s_out[6] = 0;
            s_out[7] = 0;                                  // "the misses are not in chunk-id order" (set behind barrier (G))
        }
    }
    auto insert_resident = [&]() __attribute__((always_inline)) {
        if (tid < R && my_cached >= 0) {
            unsigned pos = hash_slot(my_cached, H);
            for (int probe = 0; probe < H; ++probe) {
                int prev = atomicCAS(&s_hkeys[pos], -1, my_cached);
                if (prev == -1 || prev == my_cached) {
                    atomicMin(&s_hvals[pos], tid);
                    break;
                }
                pos = (pos + 1) & (unsigned)(H - 1);
            }
        }
    };
    int my_key = -1;   // id selected into position tid (tid < S)

    if (score != nullptr) {
        if (j0 + SEGV * 8 > N) {   // the thread(s) at the end of the row: padding -> key 0
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int j = j0 + 2 * i;
                w[i] = (j < N ? w[i] & 0xffffu : 0u) | (j + 1 < N ? w[i] & 0xffff0000u : 0u);
            }
        }
        bool placed = false;
        if constexpr (FG > 0)
            placed = t3_fused_front<FG, NW>(w, ft, lm_idx, b, N, S, j0, tid, s_hist, s_w, s_out, s_cur, s_id, s_cand, s_fin, eh,
                                            insert_resident);
        if (!placed) {
        int thr, need_eq;
        t2_find_threshold<NW>(w, T2_THREADS * SEGV * 8 - N, S, tid, s_hist, s_w, s_out, thr, need_eq, insert_resident);
        if (eh.dthr_out != nullptr && tid < eh.G) {       // next step's flag thresholds (early fetch; prediction only)
            const float kth = __uint_as_float((uint32_t)thr << 16);
            const float* fin = FG > 0 ? s_fin : eh.finals + (size_t)b * eh.G * 2;
            const float mx = fin[2 * tid], inv = fin[2 * tid + 1];
            eh.dthr_out[(size_t)b * eh.G + tid] = (kth > 0.f && inv > 0.f) ? mx + __logf(kth / inv) + eh.margin : INFINITY;
        }
        // ---- flags of the thread's keys as bit masks: bit 2i + h of word-group g <=> key (i, h) >= thr (mge) / > thr (mgt).
        // keys are < 0x8000, so key + (0x8000 - thr) has bit 15 set iff key >= thr, and the two halves of a word never carry
        // into each other
        uint32_t mge[NG], mgt[NG];
        {
            const uint32_t kge = (0x8000u - (uint32_t)thr) * 0x10001u, kgt = kge - 0x10001u;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                mge[g] = 0u;
                mgt[g] = 0u;
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const uint32_t te = w[i] + kge, tg = w[i] + kgt;
                const uint32_t fe = ((te >> 15) & 1u) | ((te >> 31) << 1), fg = ((tg >> 15) & 1u) | ((tg >> 31) << 1);
                mge[i / 16] |= fe << (2 * (i % 16));
                mgt[i / 16] |= fg << (2 * (i % 16));
            }
        }
        int cg = 0, ce = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cg += __builtin_popcount(mgt[g]);
            ce += __builtin_popcount(mge[g]);
        }
        ce -= cg;
        // slot -> chunk id of the thread's first two candidates: requested now, consumed after the scan
        long long id0 = -1, id1 = -1;
        if (lm_idx != nullptr && cg + ce > 0) {
            int ord = 0;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint32_t m = mge[g];
                if (ord < 2 && m) {
                    // (a threshold of key 0 - fewer non-zero scores than S - flags the PADDING keys of the row's tail as well:
                    // they are never placed, the tie rule stops before them, but their index lies past the row - up to 8,191
                    // for a 300-score row - so the speculative gather clamps it: round 4, a fault on an unmapped page)
                    const int e0 = __builtin_ctz(m);
                    m &= m - 1;
                    if (ord == 0) id0 = lm_idx[(size_t)b * N + min(j0 + g * 32 + e0, N - 1)];
                    else id1 = lm_idx[(size_t)b * N + min(j0 + g * 32 + e0, N - 1)];
                    ++ord;
                    if (ord < 2 && m) {
                        id1 = lm_idx[(size_t)b * N + min(j0 + g * 32 + __builtin_ctz(m), N - 1)];
                        ++ord;
                    }
                }
            }
        }
        // ---- ordered compaction: (#greater, #equal capped at the quota) before this thread, one packed scan
        ce = min(ce, need_eq);      // only "fewer than the quota so far" matters downstream; keeps the packed sum in range
        const int packed = cg | (ce << 10);
        const int pincl = block_scan_incl1(packed, s_w + 32, tid);                    // barrier (E)
        const int pexcl = pincl - packed;
        TOPK_STAMP(4);
        // A thread walks its candidates one by one, ~160 ns each (in-kernel stamps): fine for the one or two it has when the
        // selection is scattered - but hot chunks come in runs, a thread that owns a run has up to 16, and the whole workgroup
        // waits for it (+2-3 us).  A wave that holds such a thread and few enough candidates in all lets its LANES take one
        // candidate each instead: owners mark where their candidates start (LDS, the dead score histogram), a max-scan hands
        // every lane its owner, the owner's masks / prefixes come over with shuffles, and the lane places "its" candidate with
        // the same position formula as the walk below.
        const int cge = NG == 1 ? __builtin_popcount(mge[0]) : 0;
        int rounds = 0, cincl = 0;
        if (NG == 1 && __any(cge >= 4)) {                       // (uniform per wave)
            cincl = wave_scan_incl(cge);
            const int wtot = __builtin_amdgcn_readlane(cincl, 63), maxc = wave_max_i32_dpp(cge);
            // a round of 64 candidates costs about as much as three steps of the walk: worth it up to maxc / 4 rounds
            if (wtot <= 64 * min(maxc / 4, 4)) rounds = (wtot + 63) / 64;
        }
        if (rounds > 0) {
            const int lane = tid & 63, cexcl = cincl - cge, wtot = __builtin_amdgcn_readlane(cincl, 63);
            volatile int* sw = s_hist + 64 + (tid >> 6) * 64;   // this wave's marks (LDS operations of one wave execute in order)
            for (int r = 0; r < rounds; ++r) {
                const int base = 64 * r;                        // this round places candidates base .. base + 63 of the wave
                sw[lane] = 0;
                if (cge > 0 && cexcl >= base && cexcl < base + 64) sw[cexcl - base] = lane + 1;
                if (cexcl < base && cincl > base) sw[0] = lane + 1;          // the owner whose candidates straddle the round start
                const int own = wave_scan_max(sw[lane]) - 1;    // the lane that owns candidate base + `lane`
                const int src = own < 0 ? 0 : own;
                const uint32_t o_ge = (uint32_t)__shfl((int)mge[0], src, 64), o_gt = (uint32_t)__shfl((int)mgt[0], src, 64);
                const int o_pexcl = __shfl(pexcl, src, 64), o_cexcl = __shfl(cexcl, src, 64);
                const int o_id0l = __shfl((int)(id0 & 0xffffffffll), src, 64), o_id0h = __shfl((int)(id0 >> 32), src, 64);
                const int o_id1l = __shfl((int)(id1 & 0xffffffffll), src, 64), o_id1h = __shfl((int)(id1 >> 32), src, 64);
                if (base + lane < wtot) {
                    const int ord = base + lane - o_cexcl;                         // which of the owner's candidates
                    const int e = kth_set_bit(o_ge, ord);
                    const uint32_t below = (1u << e) - 1u, o_eq = o_ge & ~o_gt;
                    const int gt_run = (o_pexcl & 1023) + __builtin_popcount(o_gt & below);
                    const int eq_run = (o_pexcl >> 10) + __builtin_popcount(o_eq & below);
                    const int j = ((tid & ~63) + src) * (SEGV * 8) + e;
                    int pos = -1;
                    if ((o_gt >> e) & 1u) pos = gt_run + min(eq_run, need_eq);
                    else if (eq_run < need_eq) pos = gt_run + eq_run;
                    if (pos >= 0) {
                        const long long i0 = ((long long)o_id0h << 32) | (unsigned int)o_id0l;
                        const long long i1 = ((long long)o_id1h << 32) | (unsigned int)o_id1l;
                        s_cur[pos] = j;
                        s_id[pos] = lm_idx == nullptr ? (long long)j : ord == 0 ? i0 : ord == 1 ? i1 : -1ll;
                    }
                }
            }
        } else {
            int gt_run = pexcl & 1023, eq_run = pexcl >> 10, ord = 0;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint32_t m = mge[g];
                while (m) {
                    const int e = __builtin_ctz(m);
                    m &= m - 1;
                    int pos = -1;
                    if ((mgt[g] >> e) & 1u) {
                        pos = gt_run + min(eq_run, need_eq);
                        ++gt_run;
                    } else {
                        if (eq_run < need_eq) pos = gt_run + eq_run;
                        ++eq_run;
                    }
                    if (pos >= 0) {
                        s_cur[pos] = j0 + g * 32 + e;
                        s_id[pos] = lm_idx == nullptr ? (long long)(j0 + g * 32 + e) : ord == 0 ? id0 : ord == 1 ? id1 : -1ll;
                    }
                    ++ord;
                }
            }
        }
        }   // !placed
        __syncthreads();                                   // (F)
        TOPK_STAMP(5);
        if (tid < S) {
            long long id = s_id[tid];
            if (id < 0) id = lm_idx[(size_t)b * N + s_cur[tid]];     // third and later candidates of one thread (rare)
            my_key = (int)id;
            if (sel_out) sel_out[(size_t)b * S + tid] = id;
        }
    } else {
        if (tid < S) my_key = (int)cur_in[(size_t)b * S + tid];
        __syncthreads();     // hash arrays initialised
        insert_resident();
        __syncthreads();
    }

    // ---- classify the new ids against the resident set
#ifdef SKV_TOPK_STAMPS
    if (my_key == -12345) s_out[9] = 1;   // waits for the id gather
    TOPK_STAMP(6);
#endif
    int my_slot = -1;
    if (tid < S) {
        if (my_key >= 0) {
            unsigned pos = hash_slot(my_key, H);
            for (int probe = 0; probe < H; ++probe) {
                int k2 = s_hkeys[pos];
                if (k2 == my_key) {
                    my_slot = s_hvals[pos];
                    break;
                }
                if (k2 == -1) break;
                pos = (pos + 1) & (unsigned)(H - 1);
            }
        }
        if (my_slot >= 0) s_byslot[my_slot] = my_key;
    }
    const bool lru = R > S;                               // (uniform)
    if (lru) {
        if (tid < 64) s_age[tid] = 0;                     // all reads of the score histogram ended before barrier (D)
        const unsigned long long hb = __ballot(tid < S && my_slot >= 0);
        if ((tid & 63) == 0 && hb) atomicAdd(&s_out[6], __builtin_popcountll(hb));
    }
    __syncthreads();
    TOPK_STAMP(7);
    // hits ordered by old slot (compaction of s_byslot), misses in selection order; thread tid owns slot tid
    const int is_hit_slot = (tid < R && s_byslot[tid] >= 0) ? 1 : 0;
    const int is_miss = (tid < S && my_slot < 0) ? 1 : 0;
    // which slots the misses take: R == S every slot that is not a hit; else the S - cnt oldest of them
    int age = -1, is_gt = (tid < R && !is_hit_slot) ? 1 : 0, is_eq = 0, quota = 0;
    if (lru) {
        if (tid < R && !is_hit_slot) {
            age = my_cached < 0 ? 63 : min(max(my_age, 0), 62);
            atomicAdd(&s_age[age], 1);
        }
        __syncthreads();
        // every wave finds the age threshold itself: lane l holds the count of age 63 - l (oldest first)
        const int need = S - s_out[6], lane = tid & 63;
        const int c = s_age[63 - lane], incl = wave_scan_incl(c);
        const int first = __builtin_ctzll(__ballot(incl >= need));        // sum of all bins = R - cnt >= need
        const int thr_age = 63 - first;
        quota = need - (__builtin_amdgcn_readlane(incl, first) - __builtin_amdgcn_readlane(c, first));
        is_gt = age > thr_age ? 1 : 0;
        is_eq = age == thr_age ? 1 : 0;
    }
    int ev_pk, hm;   // (rows 48..79 of s_w: the wave maxima of the threshold search are dead)
    if (lru) {
        hm = block_scan_incl2(is_hit_slot | (is_miss << 16), is_gt | (is_eq << 16), s_w + 48, tid, ev_pk);
    } else {         // every slot that is not a hit is refilled: its rank follows from the hit prefix, one scan
        hm = block_scan_incl1(is_hit_slot | (is_miss << 16), s_w + 48, tid);
        ev_pk = tid + 1 - (hm & 0xffff);                   // inclusive count of non-hit slots up to tid (tid < R)
    }
    const int hit_incl = hm & 0xffff, miss_incl = hm >> 16;
    if (tid == T2_THREADS - 1) {
        s_out[4] = hit_incl;
        s_out[5] = miss_incl;
    }
    if (is_miss) s_miss[miss_incl - 1] = my_key;
    int* s_free = s_hvals;  // the hash values are dead after the classification: r-th slot to refill (ascending)
    const int gt_excl = (ev_pk & 0xffff) - is_gt, eq_excl = (ev_pk >> 16) - is_eq;
    const bool evict = is_gt || (is_eq && eq_excl < quota);
    if (dst_slots && evict) s_free[gt_excl + min(eq_excl, quota)] = tid;
    if (lru && tid < R)
        slot_age[(size_t)b * R + tid] = (is_hit_slot || evict) ? 0 : my_cached < 0 ? 63 : min(age + 1, 62);
    __syncthreads();
    TOPK_STAMP(8);
    const int cnt = s_out[4], nm = s_out[5];
    // misses ordered by chunk id: usually sorted already (ascending landmark slot, increasing slot -> id map).  The vote is a
    // wave ballot + one LDS flag + one barrier (__syncthreads_or is a software workgroup reduction: ~1 us at 1,024 threads,
    // in-kernel stamps)
    const int unsorted_here = (tid > 0 && tid < nm && s_miss[tid - 1] > s_miss[tid]) ? 1 : 0;
    if (__ballot(unsorted_here) != 0ull && (tid & 63) == 0) s_out[7] = 1;
    __syncthreads();
    if (s_out[7] != 0) {
        const int P = T2_THREADS / SP;
        const int i = tid / P, part = tid % P;
        if (i < nm) {
            const int ki = s_miss[i];
            int r = 0;
            for (int j = part; j < nm; j += P) {
                int kj = s_miss[j];
                r += (kj < ki) || (kj == ki && j < i);
            }
            if (r) atomicAdd(&s_rank[i], r);
        }
        __syncthreads();
    } else if (tid < nm) {
        s_rank[tid] = tid;        // read back by the same thread below
    }
    TOPK_STAMP(9);
    // ---- write out
    if (dst_slots) {
        if (tid < nm) {
            const int key = s_miss[tid], r = s_rank[tid], slot = s_free[r];
            cached[(size_t)b * R + slot] = (long long)key;
            offsets[(size_t)b * S + cnt + r] = key;
            dst_slots[(size_t)b * S + cnt + r] = slot;
        }
        if (is_hit_slot) dst_slots[(size_t)b * S + hit_incl - 1] = tid;   // [0, cnt): the slots of the hits, ascending
        if (tid == 0) cnts[b] = cnt;
        TOPK_STAMP(10);
        return;
    }
    if (is_hit_slot) {
        int o = hit_incl - 1;
        cached[(size_t)b * S + o] = (long long)s_byslot[tid];
        offsets[(size_t)b * S + o] = tid;
    }
    if (tid < nm) {
        int key = s_miss[tid], o = cnt + s_rank[tid];
        cached[(size_t)b * S + o] = (long long)key;
        offsets[(size_t)b * S + o] = key;
    }
    if (tid == 0) cnts[b] = cnt;
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
static inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

template <int G>
static int launch_score_g(const void* q, const void* lm, void* D, float* pmax, float* psum, int B, int N, int T,
                          float alpha, hipStream_t st, const EarlyHooks& eh, const FusedSel& fs) {
    hipLaunchKernelGGL((skv_score_tile_kernel<G>), dim3(T, B), dim3(64 * (G == 8 ? 8 : SKV_SCORE_WAVES)), 0, st, (const bf16_t*)q,
                       (const bf16_t*)lm, (bf16_t*)D, pmax, psum, N, T, alpha, eh, fs);
    return SKV_OK;
}

int skv_launch_score(const void* q, const void* lm, void* D, float* pmax, float* psum, int B, int G, int N,
                     float alpha, hipStream_t st, const EarlyHooks* hooks, const FusedSel* fused) {
    const int T = (N + SKV_TILE - 1) / SKV_TILE;
    EarlyHooks eh{};
    if (hooks) eh = *hooks;
    FusedSel fs{};
    if (fused) fs = *fused;
    switch (G) {
        case 1: return launch_score_g<1>(q, lm, D, pmax, psum, B, N, T, alpha, st, eh, fs);
        case 2: return launch_score_g<2>(q, lm, D, pmax, psum, B, N, T, alpha, st, eh, fs);
        case 4: return launch_score_g<4>(q, lm, D, pmax, psum, B, N, T, alpha, st, eh, fs);
        case 8: return launch_score_g<8>(q, lm, D, pmax, psum, B, N, T, alpha, st, eh, fs);
        case 16: return launch_score_g<16>(q, lm, D, pmax, psum, B, N, T, alpha, st, eh, fs);
        default: return SKV_ERR_UNSUPPORTED;
    }
}

int skv_launch_softmax_final_apply(const void* D, float* pmax, float* psum, void* P, int B, int m, int N,
                                   hipStream_t st) {
    const int T = (N + SKV_TILE - 1) / SKV_TILE;
    hipLaunchKernelGGL(skv_softmax_final_kernel, dim3(m, B), dim3(64), 0, st, pmax, psum, m, T);
    hipLaunchKernelGGL(skv_softmax_apply_kernel, dim3((N + 255) / 256, m, B), dim3(256), 0, st, (const bf16_t*)D,
                       (const float*)pmax, (const float*)psum, (bf16_t*)P, m, N, T);
    return SKV_OK;
}

int skv_launch_normalize_groupmax(const void* D, const float* pmax, const float* psum, void* P, void* score,
                                  int score_stride, int B, int G, int N, hipStream_t st, const EarlyHooks* hooks) {
    EarlyHooks eh{};
    if (hooks) eh = *hooks;
    const int T = (N + SKV_TILE - 1) / SKV_TILE;
    const int tpb = T >= 1024 ? 8 : T >= 256 ? 4 : 1;   // per-workgroup finals cost O(T): amortise them for long contexts
    const int gx = (T + tpb - 1) / tpb;
    // early fetch: one more workgroup per head (the list of chunks to pull); it is the only one that uses dynamic LDS
    const bool prep = eh.dthr_in != nullptr;
    if (prep && (T > 256 || eh.R > 1024)) return SKV_ERR_UNSUPPORTED;
    const size_t smem = prep ? skv_early_prep_lds_bytes(eh.n_chunks) : 0;
    if (smem > 60 * 1024) return SKV_ERR_UNSUPPORTED;
    const int prep_block = prep ? gx : -1;
#define SKV_NG(GG)                                                                                              \
    hipLaunchKernelGGL((skv_normalize_groupmax_kernel<GG>), dim3(gx + (prep ? 1 : 0), B), dim3(256), smem, st, (const bf16_t*)D, pmax, \
                       psum, (bf16_t*)P, (bf16_t*)score, N, T, score_stride, tpb, eh, prep_block)
    switch (G) {
        case 1: SKV_NG(1); break;
        case 2: SKV_NG(2); break;
        case 4: SKV_NG(4); break;
        case 8: SKV_NG(8); break;
        case 16: SKV_NG(16); break;
        default: return SKV_ERR_UNSUPPORTED;
    }
#undef SKV_NG
    return SKV_OK;
}

template <int SEGV, int FG>
static int launch_topk2(const void* score, int score_stride, const int64_t* lm_idx, const int64_t* cur_in, int64_t* cached,
                        int32_t* offsets, int32_t* cnts, int64_t* sel_out, int32_t* dst_slots, int B, int N, int S, int H,
                        int SP, int R, int RP, int32_t* slot_age, hipStream_t st, const EarlyHooks& eh, const FusedTop& ft) {
    const size_t smem = (size_t)(T2_BINS * T2_COPIES + SP * 5 + RP + H * 2 + 80 + 16 + (FG > 0 ? T3_CAND + 64 : 0)) * sizeof(int);
    static size_t attr_bytes[64] = {};
    if (skv_ensure_max_lds((const void*)skv_topk2_kernel<SEGV, FG>,
                           (size_t)(T2_BINS * T2_COPIES + 1024 * 6 + 4096 * 2 + 96 + (FG > 0 ? T3_CAND + 64 : 0)) * sizeof(int), attr_bytes) != SKV_OK)
        return SKV_ERR_LAUNCH;
    const bool pull = eh.staging != nullptr && eh.dthr_in != nullptr;
    EarlyHooks ek = eh;
    if (!pull) ek.staging = nullptr;
    ek.pull_wgs = B <= 8 ? SKV_EARLY_PULL_WGS : 1;
    if (pull && FG > 0 && skv_early_prep_lds_bytes(eh.n_chunks) + (EF_MAX_E + 1) * sizeof(int) > smem) return SKV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((skv_topk2_kernel<SEGV, FG>), dim3(pull ? (1 + ek.pull_wgs) * B : B), dim3(T2_THREADS), smem, st, (const bf16_t*)score, lm_idx, cur_in,
                       cached, offsets, cnts, sel_out, dst_slots, N, score_stride, S, H, SP, R, RP, slot_age, ek, ft);
    return SKV_OK;
}

// shapes the fused selection front end is instantiated for (skv_launch_select_fused)
bool skv_fused_select_supported(int G, int N, int S) {
    const int per_thread = ((N + 7) / 8 + T2_THREADS - 1) / T2_THREADS;
    return (G == 4 || G == 8) && per_thread <= 4 && (N + SKV_TILE - 1) / SKV_TILE <= 256 && S >= 1 && S <= N;
}

int skv_launch_topk_resident(const void* score, int score_stride, const int64_t* lm_idx, const int64_t* cur_in,
                            int64_t* cached, int32_t* offsets, int32_t* cnts, int64_t* sel_out, int32_t* dst_slots,
                            int B, int N, int S, int R, int32_t* slot_age, hipStream_t st, const EarlyHooks* hooks,
                            const FusedTop* fused, int G) {
    if (S < 1 || S > SKV_SEL_THREADS || R < S || R > T2_THREADS) return SKV_ERR_UNSUPPORTED;
    EarlyHooks eh{};
    if (hooks) eh = *hooks;
    if (score != nullptr && (N < S || score_stride < N || (score_stride % 8))) return SKV_ERR_ARG;
    if (R > S && (!dst_slots || !slot_age)) return SKV_ERR_ARG;      // a larger resident set exists in the in-place layout only
    const int SP = next_pow2(S), RP = next_pow2(R);
    const int H = 4 * RP;
    FusedTop ft{};
    if (fused) {                     // `score` = the scan launch's 15-bit keys; exact scores are computed from Dt in the kernel
        if (!score || !fused->Dt || !fused->part_max || !fused->part_sum || !fused->ctil || !fused->level || !skv_fused_select_supported(G, N, S))
            return SKV_ERR_UNSUPPORTED;
        ft = *fused;
        const int per_thread = (score_stride / 8 + T2_THREADS - 1) / T2_THREADS;
#define SKV_T3(SV)                                                                                                      \
    return G == 4 ? launch_topk2<SV, 4>(score, score_stride, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, B, N, S, \
                                        H, SP, R, RP, slot_age, st, eh, ft)                                             \
                  : launch_topk2<SV, 8>(score, score_stride, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, B, N, S, \
                                        H, SP, R, RP, slot_age, st, eh, ft)
        if (per_thread <= 1) { SKV_T3(1); }
        if (per_thread <= 2) { SKV_T3(2); }
        if (per_thread <= 4) { SKV_T3(4); }
#undef SKV_T3
        return SKV_ERR_UNSUPPORTED;
    }
#ifndef SKV_TOPK_V1
    {   // second-generation kernel: scores in registers, <= 16 vectors (128 scores) per thread
        const int per_thread = score ? (score_stride / 8 + T2_THREADS - 1) / T2_THREADS : 1;
        if (per_thread <= 1) return launch_topk2<1, 0>(score, score_stride, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, B, N, S, H, SP, R, RP, slot_age, st, eh, ft);
        if (per_thread <= 2) return launch_topk2<2, 0>(score, score_stride, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, B, N, S, H, SP, R, RP, slot_age, st, eh, ft);
        if (per_thread <= 4) return launch_topk2<4, 0>(score, score_stride, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, B, N, S, H, SP, R, RP, slot_age, st, eh, ft);
        if (per_thread <= 8) return launch_topk2<8, 0>(score, score_stride, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, B, N, S, H, SP, R, RP, slot_age, st, eh, ft);
        if (per_thread <= 16) return launch_topk2<16, 0>(score, score_stride, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, B, N, S, H, SP, R, RP, slot_age, st, eh, ft);
    }
#endif
    if (eh.dthr_in != nullptr) return SKV_ERR_UNSUPPORTED;   // early fetch: the pull role rides in the second-generation kernel
    if (R != S) return SKV_ERR_UNSUPPORTED;   // rows longer than 131,072 scores: first-generation kernel, R == S only
    const size_t base = (size_t)(SP * 4 + H * 2 + 256 + 32 + 8 + 256 * 32) * sizeof(int);
    const size_t with_score = base + (size_t)score_stride * sizeof(bf16_t);
    const bool stage = score != nullptr && with_score <= 150 * 1024;
    const size_t smem = stage ? with_score : base;
    if (stage) {
        static size_t attr_bytes[64] = {};
        if (smem > 64 * 1024 && skv_ensure_max_lds((const void*)skv_topk_reorder_kernel<1>, 150 * 1024, attr_bytes) != SKV_OK)
            return SKV_ERR_LAUNCH;
        hipLaunchKernelGGL(skv_topk_reorder_kernel<1>, dim3(B), dim3(SKV_SEL_THREADS), smem, st,
                           (const bf16_t*)score, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, N,
                           score_stride, S, H, SP);
    } else {
        hipLaunchKernelGGL(skv_topk_reorder_kernel<0>, dim3(B), dim3(SKV_SEL_THREADS), smem, st,
                           (const bf16_t*)score, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, N,
                           score_stride, S, H, SP);
    }
    return SKV_OK;
}

int skv_launch_topk_reorder(const void* score, int score_stride, const int64_t* lm_idx, const int64_t* cur_in,
                            int64_t* cached, int32_t* offsets, int32_t* cnts, int64_t* sel_out, int32_t* dst_slots,
                            int B, int N, int S, hipStream_t st) {
    return skv_launch_topk_resident(score, score_stride, lm_idx, cur_in, cached, offsets, cnts, sel_out, dst_slots, B, N, S, S,
                                    nullptr, st, nullptr, nullptr, 0);
}

#ifdef SKV_TOPK_STAMPS
// diagnostic builds only (libshadowkv_hip_stamps.so, tools/topk_stamps.py): the phase stamps of the last top-k launch
extern "C" __attribute__((visibility("default"))) int skv_debug_topk_stamps(unsigned long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_topk_stamps), sizeof(g_topk_stamps)) == hipSuccess ? 0 : -1;
}
#endif
