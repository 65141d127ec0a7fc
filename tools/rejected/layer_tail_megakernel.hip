// REJECTED EXPERIMENT - not part of libshadowkv_hip.so (see profiles/r03_layer_tail_megakernel.txt for the measurement and why).
// Kept as the record of what was measured; it was built as shadowkv_amd/csrc/skv_tail.hip with the C entry skv_layer_tail_bf16.
// The dense tail of one decode layer (one token, one sequence) as ONE persistent launch:
//     attention-record merge -> O projection -> residual add + RMSNorm -> gate/up projection + SiLU*mul -> down projection
// (reference: models/llama.py:354-427 post_attention_compute, models/base.py:315-341).
//
// Why one launch.  As four launches (skv_attn_merge_kernel, three skv_gemv_kernel) the 386 MB of weights stream at 6.36 TB/s
// but every launch pays ~1.9 us in front of its first byte and behind its last (measured: o_proj 33.5 MB 7.2 us, down
// 117 MB 20.4 us, gate/up 235 MB 39.1 us), the merge another 5.4 us: ~11 us of 72 with the memory pipe idle.  Here the
// weights of all three projections are ONE stream per wave: a ring of 3 batches (8 KiB of one row) per wave is always in
// flight, also across the three grid-wide dependencies (o needs every head's merge, the norm needs all of o, down needs
// all of act), so HBM keeps streaming while a barrier resolves.
//
// Structure.  One 576-thread workgroup per CU: 8 STREAMING waves and 1 HELPER wave.
//   * A streaming wave issues nothing but its ring's weight loads (vector memory returns in order per wave: a store or
//     an x load behind 24 KiB of queued weights would wait for all of them).  x comes from LDS, results go to LDS.
//   * The helper wave does everything else: the record merge (workgroups 0..Hq-1), moving the workgroup's results
//     LDS -> global, the grid barrier (release add, sc1 poll, acquire), and the next stage's x -> LDS (for gate/up: the
//     residual add + RMSNorm prologue).  It runs the 256-thread code of skv_attn_merge_kernel / skv_gemv_kernel's norm
//     prologue as 4 virtual waves, same element -> (virtual) thread mapping and reduction trees.
//   * Every row's arithmetic is the arithmetic of skv_gemv_kernel (lane l owns elements 8l..8l+7 of each 512-element
//     step, k ascending, same final tree): the outputs are bit-identical to the four launches (tests/test_gpu_graph.py).
// Grid barrier: per-XCD replicas of a monotonic 64-bit arrival counter (workgroup b adds to replica b & 7), the helper
// polls all 8 with one wave-instruction.  The poll is bounded: a launch that cannot make progress (workgroups not all
// resident) sets status != 0 and terminates; the launcher sizes the grid from the occupancy query so that cannot happen on
// an otherwise idle device, and the host refuses the fused tail when status is set.
#include "../../include/shadowkv_hip.h"
#include "skv_attn_body.h"
#include "skv_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define TL_STREAM_WAVES 8
#define TL_THREADS (64 * (TL_STREAM_WAVES + 1))
#define TL_RING 3
#define TL_MAX_REC 64                    // records per head (skv_attn_merge_kernel's MRG_MAX_REC)
#define TL_MAX_K 16384                   // x elements in LDS
#define TL_MAX_Y 1024                    // results of one workgroup and stage
#define TL_SYNC_WORDS (8 * 16 + 16)      // 8 replicas on lines of their own + status
#define TL_POLL_LIMIT (1 << 16)

// Phase stamps (diagnostic build only, -DSKV_RB_STAMPS -> libshadowkv_hip_stamps.so, tools/tail_probe.py): per workgroup, 16
// words: helper 0..9 (start, merged, barrier 0, x0 in LDS, o stored, barrier 1, x1 in LDS, act stored, barrier 2, x2 in
// LDS), streaming wave 0: 10..15 (start, stage 0 done, stage 1 started, done, stage 2 started, done).
#ifdef SKV_RB_STAMPS
__device__ unsigned long long g_tl_stamps[512 * 16];
#define TL_STAMP(i) do { if (lane == 0 && blockIdx.x < 512) g_tl_stamps[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
extern "C" __attribute__((visibility("default"))) int skv_debug_tl_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl_stamps), sizeof(g_tl_stamps)) == hipSuccess ? 0 : -1;
}
#define TL_STAMP_S(i) do { if (wave == 0) TL_STAMP(i); } while (0)
#else
#define TL_STAMP(i)
#define TL_STAMP_S(i)
#endif

struct TailArgs {
    const float* ws; const int32_t* cnts; int G, splits, tiles, Hq;        // merge
    bf16_t* attn_out;                                                      // [Hq*128] scratch (also an output)
    const bf16_t* Wo; bf16_t* o_out;                                       // [H][KA], [H] scratch
    const bf16_t* residual; const bf16_t* w_norm; float eps; bf16_t* h_out;
    const bf16_t* Wgu; int I; bf16_t* act;                                 // [2I][H], [I] scratch
    const bf16_t* Wd; bf16_t* x_out;                                       // [H][I], [H]
    unsigned long long* sync;                                              // TL_SYNC_WORDS x 8 B, zero-initialised once
    int H, KA;
};

struct TailGeom {                                                          // wave-uniform
    int NW, gw;
    int itA, itB, itC, bprC, kstepsC, tailC;
    int TA, TB, TC;                                                        // padded to multiples of TL_RING
};

__device__ __forceinline__ int tl_pad(int t) { return (t + TL_RING - 1) / TL_RING * TL_RING; }

// One batch = 8 segments of 1 KiB (64 lanes x 16 B) of ONE row: all of a 4096-element row, a quarter of a down-projection
// row.  Every load is unconditional (a batch that
// does not exist, a segment past the row and the lanes past a partial last segment read the first KiB of Wo instead, an L2
// hit): hipcc's s_waitcnt counts then stay exact, a conditional load would make every wait a vmcnt(0).
//   stage 0 (O):       unit = row,          1 batch per unit
//   stage 1 (gate/up): unit = (gate, up),   2 batches per unit (row u, row I + u)
//   stage 2 (down):    unit = row,          bprC batches per unit
template <int ST>
__device__ __forceinline__ void tl_issue(const TailArgs& a, const TailGeom& g, int s, int lane, u32x4 (&buf)[8]) {
    const bf16_t* const dummy = a.Wo + 8 * lane;
    const bf16_t* p = dummy;
    int kb = 0, segs = 0, tail = 0;
    if (ST == 0) {
        const int unit = g.gw + s * g.NW;
        if (s < g.itA && unit < a.H) { p = a.Wo + (size_t)unit * a.KA + 8 * lane; segs = 8; }
    } else if (ST == 1) {
        const int it = s >> 1, unit = g.gw + it * g.NW;
        if (it < g.itB && unit < a.I) { p = a.Wgu + ((size_t)unit + ((s & 1) ? (size_t)a.I : 0)) * a.H + 8 * lane; segs = 8; }
    } else {
        const int it = s / g.bprC, unit = g.gw + it * g.NW;
        kb = s - it * g.bprC;
        if (it < g.itC && unit < a.H) { p = a.Wd + (size_t)unit * a.I + 8 * lane; segs = g.kstepsC; tail = g.tailC; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int seg = kb * 8 + u;
        const bool on = seg < segs || (seg == segs && lane < tail);
        buf[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(on ? p + (size_t)seg * 512 : dummy));
    }
}

// issue batch s of stage ST, or - behind the stage's last batch - the next stage's batch (the stream never stops)
template <int ST>
__device__ __forceinline__ void tl_issue_from(const TailArgs& a, const TailGeom& g, int s, int lane, u32x4 (&buf)[8]) {
    const int T = ST == 0 ? g.TA : ST == 1 ? g.TB : g.TC;
    if (s < T) tl_issue<ST>(a, g, s, lane, buf);
    else if (ST < 2) tl_issue<ST + 1>(a, g, s - T, lane, buf);
}

__device__ __forceinline__ float tl_row_total(const f32x2 (&acc)[4]) {
    const float sum = ((acc[0].x + acc[0].y) + (acc[1].x + acc[1].y)) + ((acc[2].x + acc[2].y) + (acc[3].x + acc[3].y));
    return wave_tree_sum(sum);
}

template <int ST>
__device__ __forceinline__ void tl_consume(const TailArgs& a, const TailGeom& g, int s, int lane, int wave,
                                           const u32x4 (&buf)[8], f32x2 (&acc)[2][4], const u32x4* s_x, bf16_t* s_y) {
    int it, kb = 0, segs = 8, tail = 0, r = 0;
    bool first = true, last = true;
    if (ST == 0) it = s;
    else if (ST == 1) { it = s >> 1; r = s & 1; }
    else { it = s / g.bprC; kb = s - it * g.bprC; segs = g.kstepsC; tail = g.tailC; first = kb == 0; last = kb == g.bprC - 1; }
    const int nit = ST == 0 ? g.itA : ST == 1 ? g.itB : g.itC;
    const int unit = g.gw + it * g.NW;
    if (it >= nit || unit >= (ST == 1 ? a.I : a.H)) return;                // padding batch / no unit left for this wave
    f32x2 (&ac)[4] = acc[ST == 1 ? r : 0];
    if (first) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ac[j] = (f32x2){0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int seg = kb * 8 + u;
        if (seg < segs || (seg == segs && lane < tail)) {
            const u32x4 xv = s_x[seg * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 xx = (f32x2){bf_lo(xv[j]), bf_hi(xv[j])};
                const f32x2 ww = (f32x2){bf_lo(buf[u][j]), bf_hi(buf[u][j])};
                ac[j] = __builtin_elementwise_fma(ww, xx, ac[j]);
            }
        }
    }
    if (ST == 1) {
        if (r == 1) {
            const float t0 = tl_row_total(acc[0]), t1 = tl_row_total(acc[1]);
            if (lane == 0) {
                const float gt = bfr(t0), up = bfr(t1);                     // the GEMV outputs are bf16 tensors
                s_y[it * TL_STREAM_WAVES + wave] = f2bf(bfr(gt / (1.0f + __expf(-gt))) * up);
            }
        }
    } else if (last) {
        const float t0 = tl_row_total(acc[0]);
        if (lane == 0) s_y[it * TL_STREAM_WAVES + wave] = f2bf(t0);
    }
}

template <int ST>
__device__ __forceinline__ void tl_stream_stage(const TailArgs& a, const TailGeom& g, int lane, int wave,
                                                u32x4 (&ring)[TL_RING][8], f32x2 (&acc)[2][4], const u32x4* s_x,
                                                bf16_t* s_y) {
    const int T = ST == 0 ? g.TA : ST == 1 ? g.TB : g.TC;
    for (int s = 0; s < T; s += TL_RING) {
#pragma unroll
        for (int k = 0; k < TL_RING; ++k) {
            tl_consume<ST>(a, g, s + k, lane, wave, ring[k], acc, s_x, s_y);
            tl_issue_from<ST>(a, g, s + k + TL_RING, lane, ring[k]);
        }
    }
}

// ---- helper wave ---------------------------------------------------------------------------------------------------
// skv_attn_merge_kernel (skv_attn.hip) run by one wave as 4 virtual waves: same loads, weights, record order and sums.
__device__ __forceinline__ void tl_merge_head(const TailArgs& a, int bq, int lane, float* s_rec, float* s_wgt,
                                              float (*s_a)[AT_D], float* s_l) {
    const int bh = bq / a.G, nrec = a.splits + a.tiles;
    const int cnt = a.cnts[bh];
    const u32x4* src = reinterpret_cast<const u32x4*>(a.ws + (size_t)bq * nrec * AT_REC);
    const int nvec = nrec * (AT_REC / 4);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {                                 // 2 x 18 vectors per lane (64 records x 33 / 64)
        u32x4 tmp[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            const int v = lane + (half * 18 + k) * 64;
            if (v < nvec) tmp[k] = src[v];
        }
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            const int v = lane + (half * 18 + k) * 64;
            if (v < nvec) reinterpret_cast<u32x4*>(s_rec)[v] = tmp[k];
        }
    }
    const int t0 = cnt / 8;                                                // first tile with a miss chunk
    {
        const bool live = lane < nrec && (lane < a.splits || lane - a.splits >= t0);
        const float mr = live ? s_rec[lane * AT_REC + AT_D] : -INFINITY;
        const float M = wave_max_dpp(mr);
        s_wgt[lane] = (mr == -INFINITY) ? 0.f : __expf(mr - M);
    }
#pragma unroll 1
    for (int vw = 0; vw < 4; ++vw) {
        const int vt = vw * 64 + lane, d = vt & (AT_D - 1), half = vt >> 7;
        float acc = 0.f, L = 0.f;
        for (int r = half; r < nrec; r += 2) {
            const float wg = s_wgt[r];
            if (wg != 0.f) {                                               // dead records may hold anything (also NaN)
                acc = __builtin_fmaf(s_rec[r * AT_REC + d], wg, acc);
                L = __builtin_fmaf(s_rec[r * AT_REC + AT_D + 1], wg, L);
            }
        }
        s_a[half][d] = acc;
        if (d == 0) s_l[half] = L;
    }
#pragma unroll 1
    for (int vw = 0; vw < 2; ++vw) {
        const int t = vw * 64 + lane;
        a.attn_out[(size_t)bq * AT_D + t] = f2bf((s_a[0][t] + s_a[1][t]) / (s_l[0] + s_l[1]));
    }
}

// grid barrier k (0, 1, 2) of this launch: arrive (release) + bounded poll (acquire).  base: lane r < 8 holds the value
// replica r had when this launch began, rounded down to a whole launch; n_r workgroups add to replica r per barrier.
__device__ __forceinline__ void tl_grid_barrier(unsigned long long* sync, int k, int lane, unsigned long long base, int n_r) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the helper's own stores (results, merge, h_out)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0)
        __hip_atomic_fetch_add(sync + (size_t)(blockIdx.x & 7) * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = base + (unsigned long long)(k + 1) * n_r;
    int polls = 0;
    for (;;) {
        unsigned long long c = target;
        if (lane < 8) c = __hip_atomic_load(sync + (size_t)lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_read_exec() == __builtin_amdgcn_ballot_w64(c >= target)) break;
        if (++polls > TL_POLL_LIMIT) {                                     // cannot happen with every workgroup resident
            if (lane == 0) atomicExch(reinterpret_cast<unsigned int*>(sync + 8 * 16), 1u);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(TL_THREADS) void skv_layer_tail_kernel(const TailArgs a) {
    __shared__ __attribute__((aligned(16))) float s_big[TL_MAX_REC * AT_REC];     // merge records, then x (<= 32 KiB)
    __shared__ __attribute__((aligned(16))) bf16_t s_y[TL_MAX_Y];
    __shared__ float s_wgt[TL_MAX_REC];
    __shared__ float s_a[2][AT_D];
    __shared__ float s_l[2];
    static_assert(TL_MAX_REC * AT_REC * 4 >= TL_MAX_K * 2, "x must fit the record area");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4* const s_x = reinterpret_cast<u32x4*>(s_big);

    TailGeom g;
    g.NW = gridDim.x * TL_STREAM_WAVES;
    g.gw = blockIdx.x * TL_STREAM_WAVES + wave;
    g.itA = (a.H + g.NW - 1) / g.NW;
    g.itB = (a.I + g.NW - 1) / g.NW;
    g.itC = g.itA;
    g.kstepsC = a.I / 512;
    g.tailC = (a.I % 512) / 8;
    g.bprC = (g.kstepsC + (g.tailC ? 1 : 0) + 7) / 8;
    g.TA = tl_pad(g.itA); g.TB = tl_pad(g.itB * 2); g.TC = tl_pad(g.itC * g.bprC);

    if (wave < TL_STREAM_WAVES) {
        // ------------------------------------------------------------------ streaming waves
        u32x4 ring[TL_RING][8];
        f32x2 acc[2][4];
#pragma unroll
        for (int k = 0; k < TL_RING; ++k) tl_issue_from<0>(a, g, k, lane, ring[k]);
        TL_STAMP_S(10);
        __syncthreads();                                                   // (x0) merged attention output in LDS
        tl_stream_stage<0>(a, g, lane, wave, ring, acc, s_x, s_y);
        TL_STAMP_S(11);
        __syncthreads();                                                   // (y0) this workgroup's o rows in LDS
        __syncthreads();                                                   // (x1) normed hidden state in LDS
        TL_STAMP_S(12);
        tl_stream_stage<1>(a, g, lane, wave, ring, acc, s_x, s_y);
        TL_STAMP_S(13);
        __syncthreads();                                                   // (y1)
        __syncthreads();                                                   // (x2) act in LDS
        TL_STAMP_S(14);
        tl_stream_stage<2>(a, g, lane, wave, ring, acc, s_x, s_y);
        TL_STAMP_S(15);
        __syncthreads();                                                   // (y2)
        return;
    }
    // ---------------------------------------------------------------------- helper wave
    unsigned long long* const sync = a.sync;
    const int r_mine = lane & 7;
    const int n_r = ((int)gridDim.x - r_mine + 7) / 8;                      // workgroups b with (b & 7) == r
    unsigned long long base = 0;
    if (lane < 8) {
        const unsigned long long c = __hip_atomic_load(sync + (size_t)lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = c - c % (3ull * n_r);                                        // (nobody can be past barrier 0 before we arrive)
    }
    // stage 0: merge the attention records of head blockIdx.x
    TL_STAMP(0);
    if ((int)blockIdx.x < a.Hq) tl_merge_head(a, blockIdx.x, lane, s_big, s_wgt, s_a, s_l);
    TL_STAMP(1);
    tl_grid_barrier(sync, 0, lane, base, n_r);
    TL_STAMP(2);
    // x0 = attention output [KA]
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.attn_out);
        const int nvec = a.KA / 8;
        for (int v0 = 0; v0 < nvec; v0 += 64 * 8) {
            u32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (v0 + k * 64 + lane < nvec) t[k] = src[v0 + k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (v0 + k * 64 + lane < nvec) s_x[v0 + k * 64 + lane] = t[k];
        }
    }
    TL_STAMP(3);
    __syncthreads();                                                       // (x0)
    __syncthreads();                                                       // (y0)
    {   // o rows of this workgroup: rows b*8 + w (+ it*NW): 8 contiguous values per iteration
        const int n = g.itA * TL_STREAM_WAVES;
        for (int i = lane; i < n; i += 64) {
            const int it = i / TL_STREAM_WAVES, l = i % TL_STREAM_WAVES;
            const int row = (int)blockIdx.x * TL_STREAM_WAVES + it * g.NW + l;
            if (row < a.H) a.o_out[row] = s_y[i];
        }
    }
    TL_STAMP(4);
    tl_grid_barrier(sync, 1, lane, base, n_r);
    TL_STAMP(5);
    {   // x1: h = o + residual, RMS statistics, xn = h * rstd * w_norm (skv_gemv_kernel's NORM prologue, 4 virtual waves)
        u32x4 hx[4][2];
        float ssv[4];
#pragma unroll
        for (int vw = 0; vw < 4; ++vw) {
            float ss = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int v = vw * 64 + lane + it * 256;
                u32x4 x = reinterpret_cast<const u32x4*>(a.o_out)[v];
                if (a.residual) {
                    const u32x4 c = reinterpret_cast<const u32x4*>(a.residual)[v];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = pack_bf2(bf_lo(x[j]) + bf_lo(c[j]), bf_hi(x[j]) + bf_hi(c[j]));
                }
                hx[vw][it] = x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ss = __builtin_fmaf(bf_lo(x[j]), bf_lo(x[j]), ss);
                    ss = __builtin_fmaf(bf_hi(x[j]), bf_hi(x[j]), ss);
                }
                if (a.h_out && blockIdx.x == 0) reinterpret_cast<u32x4*>(a.h_out)[v] = x;
            }
            ssv[vw] = wave_tree_sum(ss);
        }
        const float tot = (ssv[0] + ssv[1]) + (ssv[2] + ssv[3]);
        const float rstd = 1.0f / sqrtf(tot / (float)a.H + a.eps);
#pragma unroll
        for (int vw = 0; vw < 4; ++vw)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int v = vw * 64 + lane + it * 256;
                const u32x4 gw = reinterpret_cast<const u32x4*>(a.w_norm)[v];
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = pack_bf2(bf_lo(hx[vw][it][j]) * rstd * bf_lo(gw[j]), bf_hi(hx[vw][it][j]) * rstd * bf_hi(gw[j]));
                s_x[v] = o;
            }
    }
    TL_STAMP(6);
    __syncthreads();                                                       // (x1)
    __syncthreads();                                                       // (y1)
    {   // act values of this workgroup: 8 contiguous per iteration
        const int n = g.itB * TL_STREAM_WAVES;
        for (int i = lane; i < n; i += 64) {
            const int it = i / TL_STREAM_WAVES, l = i % TL_STREAM_WAVES;
            const int idx = (int)blockIdx.x * TL_STREAM_WAVES + it * g.NW + l;
            if (idx < a.I) a.act[idx] = s_y[i];
        }
    }
    TL_STAMP(7);
    tl_grid_barrier(sync, 2, lane, base, n_r);
    TL_STAMP(8);
    {   // x2 = act [I]
        const u32x4* src = reinterpret_cast<const u32x4*>(a.act);
        const int nvec = a.I / 8;
        for (int v0 = 0; v0 < nvec; v0 += 64 * 8) {
            u32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (v0 + k * 64 + lane < nvec) t[k] = src[v0 + k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (v0 + k * 64 + lane < nvec) s_x[v0 + k * 64 + lane] = t[k];
        }
    }
    TL_STAMP(9);
    __syncthreads();                                                       // (x2)
    __syncthreads();                                                       // (y2)
    {
        const int n = g.itC * TL_STREAM_WAVES;
        for (int i = lane; i < n; i += 64) {
            const int it = i / TL_STREAM_WAVES, l = i % TL_STREAM_WAVES;
            const int row = (int)blockIdx.x * TL_STREAM_WAVES + it * g.NW + l;
            if (row < a.H) a.x_out[row] = s_y[i];
        }
    }
}

extern "C" size_t skv_layer_tail_workspace_bytes(void) { return (size_t)TL_SYNC_WORDS * 8; }

static int tail_grid(int* grid_out) {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SKV_ERR_LAUNCH;
    if (!cached[dev]) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return SKV_ERR_LAUNCH;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, skv_layer_tail_kernel, TL_THREADS, 0) != hipSuccess)
            return SKV_ERR_LAUNCH;
        cached[dev] = (per_cu >= 1 && cus >= 8) ? cus : -1;                // one workgroup per CU: all of them resident
    }
    if (cached[dev] < 0) return SKV_ERR_UNSUPPORTED;
    *grid_out = cached[dev];
    return SKV_OK;
}

extern "C" int skv_layer_tail_bf16(const void* attn_workspace, const int32_t* cnts, int q_heads, int kv_heads, int select_sets,
                                   int attn_splits, void* attn_out, const void* Wo, void* o_out, const void* residual,
                                   const void* norm_weight, float eps, void* h_out, const void* Wgu, int intermediate,
                                   void* act, const void* Wdown, void* x_out, int hidden, void* sync_workspace,
                                   skv_stream_t stream) {
    if (!attn_workspace || !cnts || !attn_out || !Wo || !o_out || !norm_weight || !Wgu || !act || !Wdown || !x_out ||
        !sync_workspace)
        return SKV_ERR_ARG;
    if (kv_heads < 1 || q_heads % kv_heads || select_sets < 8 || select_sets % 8 || attn_splits < 1) return SKV_ERR_ARG;
    if (attn_splits + select_sets / 8 > TL_MAX_REC) return SKV_ERR_UNSUPPORTED;
    const int KA = q_heads * AT_D;
    if (hidden != 4096 || KA != 4096 || intermediate % 8 || intermediate < 512 || intermediate > TL_MAX_K)
        return SKV_ERR_UNSUPPORTED;                                        // (the norm prologue's 256 x 16 element mapping)
    int grid = 0;
    const int rc = tail_grid(&grid);
    if (rc != SKV_OK) return rc;
    if (grid < q_heads) return SKV_ERR_UNSUPPORTED;
    const int NW = grid * TL_STREAM_WAVES;
    const int itA = (hidden + NW - 1) / NW, itB = (intermediate + NW - 1) / NW;
    if (itA * TL_STREAM_WAVES > TL_MAX_Y || itB * TL_STREAM_WAVES > TL_MAX_Y) return SKV_ERR_UNSUPPORTED;
    TailArgs a{(const float*)attn_workspace, cnts, q_heads / kv_heads, attn_splits, select_sets / 8, q_heads,
               (bf16_t*)attn_out, (const bf16_t*)Wo, (bf16_t*)o_out, (const bf16_t*)residual, (const bf16_t*)norm_weight, eps,
               (bf16_t*)h_out, (const bf16_t*)Wgu, intermediate, (bf16_t*)act, (const bf16_t*)Wdown, (bf16_t*)x_out,
               (unsigned long long*)sync_workspace, hidden, KA};
    hipLaunchKernelGGL(skv_layer_tail_kernel, dim3(grid), dim3(TL_THREADS), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

extern "C" int skv_layer_tail_status(const void* sync_workspace, int* status_out) {
    if (!sync_workspace || !status_out) return SKV_ERR_ARG;
    unsigned long long v = 0;
    if (hipMemcpy(&v, (const unsigned long long*)sync_workspace + 8 * 16, 8, hipMemcpyDeviceToHost) != hipSuccess)
        return SKV_ERR_LAUNCH;
    *status_out = (int)(v & 0xffffffffull);
    return SKV_OK;
}
