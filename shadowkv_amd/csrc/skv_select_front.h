// Front end of the exact top-k selection shared by the chunk-selection kernel (skv_select.hip) and the sampler
// (skv_sample.hip): DPP scans, the one-barrier block scan, packed 16-bit helpers and the histogram search for the exact
// k-th largest 16-bit key of a row held in registers.  See skv_topk2_kernel for the design notes.
#pragma once
#include "skv_common.h"

// Phase stamps for tools/topk_probe.hip (diagnostic build only, -DSKV_TOPK_STAMPS; no stamp executes in the
// shipped library).  100 MHz wall clock, written by thread 0 of workgroup 0 to a buffer nothing else reads.
#ifdef SKV_TOPK_STAMPS
__device__ unsigned long long g_topk_stamps[32];
#define TOPK_STAMP(i)                                                                 \
    do {                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_topk_stamps[i] = wall_clock64();   \
    } while (0)
// the FIRST pull workgroup of a launch (early fetch): stamps 24 ..
#define PULL_STAMP(i)                                                                 \
    do {                                                                              \
        if (pull_stamp_wg && threadIdx.x == 0) g_topk_stamps[i] = wall_clock64();     \
    } while (0)
#else
#define TOPK_STAMP(i)
#define PULL_STAMP(i)
#endif

// Inclusive integer scans on DPP (no LDS crossbar round trips): within rows of 16 lanes row_shr 1/2/4/8, then
// row_bcast15 into rows 1 and 3 and row_bcast31 into rows 2-3 (the GFX9 wave64 scan).  Lanes without a source read
// `old` = 0.
__device__ __forceinline__ int row16_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    return v;
}
__device__ __forceinline__ int wave_scan_incl(int v) {
    v = row16_scan_incl(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// 15-bit monotone key of kappa (x <= y  =>  key(x) <= key(y)): fixed point, 1/256 per code over [-128, 0) - floor((x + 128) 256),
// clamped to [0, 32767]; NaN -> 0.  (Round 4, first version: sign / exponent / 6 mantissa bits of the f32 pattern - at kappa ~ -10,
// GLM-4's 25 K landmarks per head, one code was 0.125 wide = ~250 slots, and the witness level below could only move in steps of
// 250 slots: it alternated between too few and too many.)  skv_kappa_key_low(k) is the SMALLEST value carrying key k (for k > 0;
// key 0 also holds everything below -128), so "key(x) < k" implies "x < skv_kappa_key_low(k)" and "at least S keys >= k" implies
// "at least S values >= skv_kappa_key_low(k)" - all the fused selection's exactness argument needs from the quantisation.
#define SKV_KAPPA_SCALE 256.0f
#define SKV_KAPPA_BIAS 128.0f
__host__ __device__ __forceinline__ uint16_t skv_kappa_key(float x) {
    if (!(x > -SKV_KAPPA_BIAS)) return 0;                     // (also NaN)
    const float t = (x + SKV_KAPPA_BIAS) * SKV_KAPPA_SCALE;   // exact scaling; the sum rounds to nearest, monotone
    return (uint16_t)(t >= 32767.0f ? 32767 : (int)t);
}
__host__ __device__ __forceinline__ float skv_kappa_key_low(int key) {
    // one code below the key's nominal edge: covers the rounding of (x + 128) in skv_kappa_key (x within 2^-17 of an edge
    // may land on either side; 2^-8 >> that)
    return key <= 0 ? -INFINITY : ((float)(key - 1)) * (1.0f / SKV_KAPPA_SCALE) - SKV_KAPPA_BIAS;
}
#define T3_CAND 2048                    // candidates the fused selection evaluates exactly on its fast path (two per thread)

#define T2_THREADS 1024
#define T2_BINS 4096
#define T2_COPIES 4

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b));
}

__device__ __forceinline__ int wave_max_i32_dpp(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// inclusive scan over the 1,024-thread workgroup, ONE barrier: each wave publishes its total, then re-scans the 16
// totals on its own lanes 0..15 (DPP row scan) and picks its prefix with a readlane.  s_w[16] must not be rewritten
// before another barrier.
__device__ __forceinline__ int block_scan_incl1(int v, int* s_w, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    v = wave_scan_incl(v);
    if (lane == 63) s_w[wave] = v;
    __syncthreads();
    int w = lane < 16 ? s_w[lane] : 0;
    w = row16_scan_incl(w);
    const int pre = wave > 0 ? __builtin_amdgcn_readlane(w, wave - 1) : 0;
    return v + pre;
}

// inclusive MAX scan over the wave (values >= 0), same DPP ladder as wave_scan_incl
__device__ __forceinline__ int wave_scan_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return v;
}

// position of the k-th (0-based) set bit of m, k < popcount(m): five halvings
__device__ __forceinline__ int kth_set_bit(uint32_t m, int k) {
    int pos = 0;
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        const uint32_t low = m & ((1u << w) - 1u);
        const int c = __builtin_popcount(low);
        const bool up = k >= c;
        k -= up ? c : 0;
        m = up ? m >> w : low;
        pos += up ? w : 0;
    }
    return pos;
}

// two inclusive scans behind ONE barrier (s_w: two rows of 16); returns the first, the second through v2_incl
__device__ __forceinline__ int block_scan_incl2(int v1, int v2, int* s_w, int tid, int& v2_incl) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    v1 = wave_scan_incl(v1);
    v2 = wave_scan_incl(v2);
    if (lane == 63) {
        s_w[wave] = v1;
        s_w[16 + wave] = v2;
    }
    __syncthreads();
    int w1 = lane < 16 ? s_w[lane] : 0, w2 = lane < 16 ? s_w[16 + lane] : 0;
    w1 = row16_scan_incl(w1);
    w2 = row16_scan_incl(w2);
    const int p1 = wave > 0 ? __builtin_amdgcn_readlane(w1, wave - 1) : 0;
    const int p2 = wave > 0 ? __builtin_amdgcn_readlane(w2, wave - 1) : 0;
    v2_incl = v2 + p2;
    return v1 + p1;
}

// Home slot of a chunk id in the resident-set hash (open addressing, linear probing, H a power of two).  Multiplicative
// (Fibonacci) hashing: selected chunks come in runs of consecutive ids (neighbouring chunks are attended together), and with
// id & (H - 1) a run of 100 ids is one 100-slot cluster that every colliding insert / lookup walks with an LDS atomic per
// step (measured: the kernel went from 8.5 to 29 us when 102 consecutive ids were resident).
__device__ __forceinline__ unsigned hash_slot(int id, int H) {
    return ((unsigned)id * 2654435761u) >> (32 - (31 - __builtin_clz((unsigned)H)));
}

// Exact k-th largest key (and the tie quota) of the row whose keys the 1,024 threads hold in w[NW] (two 16-bit keys per
// word; padding already rewritten to key 0, n_pad of them).  thr = the S-th largest key, need_eq = how many keys equal to
// thr belong to the top S.  round0_extra() runs between the first histogram pass and its barrier (LDS work that overlaps).
// s_hist [T2_BINS][T2_COPIES] zeroed by the caller before its first barrier; s_w [80], s_out [16] scratch.
// need2 > S (optional, prediction only - the near-miss lists of skv_select.hip): thr2[0] = the need2-th largest key, thr2[1] = the
// (2 need2 - S)-th, when they lie in the first histogram window (else, or with fewer keys, 0 = "every key"); costs two more
// compares per thread.
template <int NW, typename F>
__device__ __forceinline__ void t2_find_threshold(const uint32_t (&w)[NW], const int n_pad, const int S, const int tid,
                                                  int* s_hist, int* s_w, int* s_out, int& thr, int& need_eq,
                                                  F round0_extra, const int need2 = 0, int* thr2 = nullptr) {
    const int lane = tid & 63, wave = tid >> 6;
    if (need2 > 0 && tid == 0) s_out[2] = s_out[3] = T2_BINS - 1;     // (visible behind barrier (A); rewritten between (C) and (D))
    {
        uint32_t m2 = w[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) m2 = pk_max_u16(m2, w[i]);
        int kmx = wave_max_i32_dpp((int)max(m2 & 0xffffu, m2 >> 16));
        if (lane == 0) s_w[64 + wave] = kmx;
        __syncthreads();                                   // (A) LDS initialised, wave maxima visible
        TOPK_STAMP(1);
        int base;
        {
            int wm = lane < 16 ? s_w[64 + lane] : 0;
            base = wave_max_i32_dpp(wm);
        }
        char* const hb = reinterpret_cast<char*>(s_hist) + (lane & (T2_COPIES - 1)) * 4;   // this lane's copy
        int need = S;
        thr = 0;
        need_eq = 0;
        for (int round = 0;; ++round) {
            // keys in (base - 4095, base] get their own bin (rel = base - key), everything lower shares bin 4095
            if (round == 0) {      // base is the maximum: every key is <= base
                const uint32_t base2 = (uint32_t)base * 0x10001u, cap2 = (uint32_t)(T2_BINS - 1) * 0x10001u;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const uint32_t r2 = pk_min_u16(pk_sub_u16(base2, w[i]), cap2);
                    atomicAdd(reinterpret_cast<int*>(hb + ((r2 & 0xffffu) << 4)), 1);
                    atomicAdd(reinterpret_cast<int*>(hb + ((r2 >> 16) << 4)), 1);
                }
                round0_extra();
            } else {
#pragma unroll
                for (int i = 0; i < NW; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int val = h ? (int)(w[i] >> 16) : (int)(w[i] & 0xffffu);
                        if (val <= base) atomicAdd(reinterpret_cast<int*>(hb + (min(base - val, T2_BINS - 1) << 4)), 1);
                    }
            }
            __syncthreads();                               // (B)
            TOPK_STAMP(2);
            // fold the copies: thread t owns bins 4t .. 4t+3 (ascending rel = descending key)
            int c[4];
            {
                const u32x4* hw = reinterpret_cast<const u32x4*>(s_hist + (size_t)tid * 4 * T2_COPIES);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const u32x4 w4 = hw[x];
                    c[x] = (int)((w4[0] + w4[1]) + (w4[2] + w4[3]));
                }
                const int relz = min(base, T2_BINS - 1);   // where the padding zeros were counted
                if ((relz >> 2) == tid) {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if ((relz & 3) == x) c[x] -= n_pad;
                }
            }
            const int tot = (c[0] + c[1]) + (c[2] + c[3]);
            const int incl = block_scan_incl1(tot, s_w + 16 * (round & 1), tid);      // barrier (C)
            int run = incl - tot;
            if (need2 > 0 && round == 0) {
                const int need3 = 2 * need2 - need;              // (need == S in round 0)
                if (run < need3 && incl >= need2) {
                    int r2 = run;
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        if (r2 < need2 && r2 + c[x] >= need2) s_out[2] = 4 * tid + x;
                        if (r2 < need3 && r2 + c[x] >= need3) s_out[3] = 4 * tid + x;
                        r2 += c[x];
                    }
                }
            }
            if (run < need && incl >= need) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    if (run < need && run + c[x] >= need) {
                        s_out[0] = 4 * tid + x;
                        s_out[1] = run;
                    }
                    run += c[x];
                }
            }
            __syncthreads();                               // (D)
            TOPK_STAMP(3);
            const int rel_thr = s_out[0], above = s_out[1];
            if (need2 > 0 && round == 0) {
                thr2[0] = s_out[2] < T2_BINS - 1 ? base - s_out[2] : 0;
                thr2[1] = s_out[3] < T2_BINS - 1 ? base - s_out[3] : 0;
            }
            if (rel_thr < T2_BINS - 1) {
                thr = base - rel_thr;
                need_eq = need - above;
                break;
            }
            // the k-th value lies among the keys <= base - 4095: slide the window (never for softmax scores of one head)
            need -= above;
            base -= T2_BINS - 1;
            u32x4* hz = reinterpret_cast<u32x4*>(s_hist);
#pragma unroll
            for (int k = 0; k < T2_BINS * T2_COPIES / 4 / T2_THREADS; ++k) hz[tid + k * T2_THREADS] = (u32x4){0u, 0u, 0u, 0u};
            __syncthreads();
        }
    }
}
