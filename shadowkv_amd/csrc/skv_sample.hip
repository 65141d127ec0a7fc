// End of the decode step (LLM.generate's loop body, /root/reference/models/base.py:628-635, sampling semantics of
// models/tensor_op.py:242-297 sample_token): top-p filter over the k sorted top-k logits, multinomial draw, and the
// step's device-side bookkeeping (RoPE position, generated-row slot, attended length, query-table index) in ONE launch.
// The PyTorch formulation is ~20 tiny launches (softmax, cumsum, cat, masked_fill, softmax, exponential_, div, argmax,
// gather, copy, and 2-3 per counter), ~60 us of a 4.9 ms step at bs = 1.
//   vals [bs][k] f32: top-k logits / temperature, sorted descending (torch.topk);  idx [bs][k] int64 their token ids
//   keep i  <=>  i == 0 or cumsum(softmax(vals))[i-1] <= top_p        (tensor_op.py:258-266 after the top-k filter)
//   draw     =  argmax_i p_i / e_i,  e_i ~ Exp(1)                     (== multinomial(p, 1))
// The uniforms come from a counter-based hash of (seed, position of the sequence, batch row, i): reproducible, no
// generator state, graph-capturable.  One wave per sequence; k <= 64.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"
#include "skv_select_front.h"

__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // lowbias32 finaliser
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(64) void skv_sample_advance_kernel(
    const float* __restrict__ vals, const int64_t* __restrict__ idx, int k, float top_p, unsigned long long seed,
    int64_t* __restrict__ token /*[bs]*/, int64_t* __restrict__ pos /*[bs]*/, int64_t* __restrict__ gen /*[1]*/,
    int64_t* __restrict__ row_idx /*[1]*/, int32_t* __restrict__ kv_len /*[1]*/, int64_t* __restrict__ step_idx /*[1]*/,
    long long base, long long slack, long long table_len, const int32_t* __restrict__ hit_cnts, int n_hit_cnts,
    int64_t* __restrict__ hit_accum) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const long long p0 = pos[b];
    const float v = lane < k ? vals[(size_t)b * k + lane] : -INFINITY;
    const float mx = wave_max_dpp(v);
    float e = lane < k ? __expf(v - mx) : 0.f;
    const float tot = wave_tree_sum(e);
    const float p = e / tot;
    // inclusive scan of p over the lanes (descending probabilities), exclusive = cumsum[i-1]
    float c = p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(c, o, 64);
        if (lane >= o) c += n;
    }
    const bool keep = lane < k && (top_p <= 0.f || lane == 0 || (c - p) <= top_p);
    const float pk = keep ? e : 0.f;                     // renormalisation is a common factor: irrelevant to the argmax
    // u in (0, 1]: 24 random bits
    uint32_t h = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9e3779b9u * (uint32_t)p0));
    h = mix32(h ^ (0x85ebca6bu * (uint32_t)(b + 1)) ^ (0xc2b2ae35u * (uint32_t)(lane + 1)) ^ (uint32_t)(p0 >> 32));
    const float u = ((h >> 8) + 1) * (1.0f / 16777216.0f);
    const float ex = -__logf(u);                          // Exp(1), > 0 except u == 1 -> 0: guarded below
    float score = keep ? pk / fmaxf(ex, 1e-30f) : -1.f;
    // argmax with lowest-lane tie break
    float best = wave_max_dpp(score);
    const unsigned long long m = __ballot(score == best);
    const int win = __ffsll((long long)m) - 1;
    if (lane == win) token[b] = idx[(size_t)b * k + lane];
    if (b == 0 && hit_accum != nullptr) {                 // statistics: chunk hits of this step (all layers' cnts)
        int hsum = 0;
        for (int i = lane; i < n_hit_cnts; i += 64) hsum += hit_cnts[i];
        hsum = wave_sum_i32(hsum);
        if (lane == 0) hit_accum[0] += hsum;
    }
    if (lane == 0) {
        pos[b] = p0 + 1;
        if (b == 0) {                                     // gen counts generated tokens; past the slack the rows form a
            const long long g2 = gen[0] + 1;              // ring of the last `slack` tokens (GraphDecoder: benchmarks only,
            gen[0] = g2;                                  // the host refuses to step there otherwise)
            row_idx[0] = base + g2 % slack;
            kv_len[0] = (int32_t)(base + (g2 + 1 < slack ? g2 + 1 : slack));
            if (step_idx) step_idx[0] = (step_idx[0] + 1) % table_len;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The whole end of step in ONE launch, straight from the bf16 logits of the lm_head: exact top-k (k <= 64) of the V
// logits of a sequence, temperature, top-p, draw, counters.  Replaces torch.topk over [bs, 128256] (83 us sbtopk +
// radix-sort / partition launches per token at bs = 1; its multi-block path faulted under hipGraph replay for bs > 1) +
// the f32 conversion and division of the whole logit row.  The selection is the front end of the chunk-selection kernel
// (skv_select_front.h): the row lives in registers, one 4,096-bin histogram pass finds the k-th largest value, an
// ordered compaction collects the winners: every logit above the k-th largest value and ALL logits equal to it (the
// reference's filter removes only logits < the k-th value, models/tensor_op.py:253-255), 64 winners at most - beyond
// that the lowest token ids among the tied ones.
// Signed bf16 logits are mapped to order-preserving unsigned 16-bit keys first.  One 1,024-thread workgroup per sequence.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bf16x2_to_keys(uint32_t w) {       // x >= 0: x | 0x8000;  x < 0: ~x
    const uint32_t neg = (w >> 15) & 0x10001u;
    return w ^ ((neg * 0xffffu) | 0x80008000u);
}
__device__ __forceinline__ float key_to_float(int key) {
    const uint32_t x = (key & 0x8000) ? (uint32_t)(key ^ 0x8000) : (uint32_t)(~key & 0xffff);
    return __uint_as_float(x << 16);
}

#define SMP_CAND (2 * T2_THREADS)    // candidates the exact search takes (two per thread)
#define SMP_KEEP 64                  // winners a row (or a part of it) can hand on: k plus the logits tied with the k-th
#define SMP_PARTS 4                  // a row is searched in parts of <= 131,072 logits (16 vectors per thread)

// Exact top-k among C <= SMP_CAND candidate logits (two per thread, candidate c = local token id id_of(c), ascending): the
// k-th largest value by the histogram search, then every candidate above it and the ones equal to it (ascending id, up to
// SMP_KEEP winners in all) into s_cur[0 .. n) with their keys in s_cur[SMP_KEEP ..]; returns n.  The histogram must be zero;
// the caller ends with a barrier.
template <typename F>
__device__ __forceinline__ int sample_candidates_topk(const bf16_t* __restrict__ row, const int C, const int k, const int tid,
                                                      int* s_hist, int* s_w, int* s_out, int* s_cur, F id_of) {
        const int c0 = 2 * tid, c1 = 2 * tid + 1;
        const int id0 = c0 < C ? id_of(c0) : -1, id1 = c1 < C ? id_of(c1) : -1;
        const uint32_t k0 = bf16x2_to_keys((uint32_t)row[max(id0, 0)]) & 0xffffu, k1 = bf16x2_to_keys((uint32_t)row[max(id1, 0)]) & 0xffffu;
        const uint32_t w2[1] = {(id0 >= 0 ? k0 : 0u) | ((id1 >= 0 ? k1 : 0u) << 16)};
        int thr, need_eq;
        t2_find_threshold<1>(w2, 2 * T2_THREADS - C, k, tid, s_hist, s_w, s_out, thr, need_eq, [] {});
        TOPK_STAMP(23);
        const int lo = (int)(w2[0] & 0xffffu), hi = (int)(w2[0] >> 16);
        const bool v0 = id0 >= 0, v1 = id1 >= 0;
        const int g0 = v0 && lo > thr, g1 = v1 && hi > thr, e0 = v0 && lo == thr, e1 = v1 && hi == thr;
        const int packed = (g0 + g1) | ((e0 + e1) << 10);             // (<= 64 greater, <= 2,048 equal in all)
        const int pincl = block_scan_incl1(packed, s_w + 48, tid);
        if (tid == T2_THREADS - 1) s_out[9] = pincl >> 10;            // logits equal to the k-th value
        __syncthreads();
        const int n_gt = k - need_eq, keep_eq = min(s_out[9], SMP_KEEP - n_gt);
        const int pexcl = pincl - packed;
        int gt_run = pexcl & 1023, eq_run = pexcl >> 10;
        // (the winner's key travels with its id - s_cur[SMP_KEEP + slot] -: no read-back of the logit)
        auto put = [&](int slot, int idv, int keyv) __attribute__((always_inline)) {
            s_cur[slot] = idv;
            s_cur[SMP_KEEP + slot] = keyv;
        };
        if (g0) put(gt_run++ + min(eq_run, keep_eq), id0, lo);
        else if (e0) { if (eq_run < keep_eq) put(gt_run + eq_run, id0, lo); ++eq_run; }
        if (g1) put(gt_run + min(eq_run, keep_eq), id1, hi);
        else if (e1 && eq_run < keep_eq) put(gt_run + eq_run, id1, hi);
        return n_gt + keep_eq;
}

// The same through RANGE MAXIMA (round 4): range_max[r] = key of the largest of logits 16 r .. 16 r + 15, left by the lm_head
// launch (skv_gemv.hip, RMAX).  The k-th largest range maximum L is a lower bound of the k-th largest logit (k ranges hold a
// logit >= L), so every winner lies in a range whose maximum reaches L: about k ranges = 16 k candidates.  The workgroup
// reads n_ranges keys (16 KB at 128 K logits) and ~1,000 logits instead of streaming the whole row through ONE CU (256 KB
// at the per-CU rate: 10.5 of the launch's 23 us, profiles/r04_sample_stamps.txt).  Returns the number of winners in
// s_cur (as sample_part_topk, local id = token id), or -1 when more than SMP_CAND / 16 ranges qualify (thousands of tied
// logits: the caller streams the row).  n_ranges <= 16,384, n_ranges >= k.  Ends with a barrier unless it returns -1.
#define SMP_RANGE 16
__device__ __forceinline__ int sample_ranges_topk(const bf16_t* __restrict__ row, const uint16_t* __restrict__ rmax, const int n_ranges,
                                                  const int k, const int tid, int* s_hist, int* s_w, int* s_out, int* s_cidx,
                                                  int* s_cur) {
    constexpr int SEGR = 2, NW = 4 * SEGR;                              // 16 range keys per thread
    uint32_t w[NW];
    const int r0 = tid * NW * 2;
    TOPK_STAMP(19);
    {
        const u32x4* gvec = reinterpret_cast<const u32x4*>(rmax);
        const int nvec = n_ranges / 8;                                  // (whole vectors; the tail below)
#pragma unroll
        for (int q = 0; q < SEGR; ++q) {
            const int vi = tid * SEGR + q;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (vi < nvec) v = gvec[vi];
            else if (vi == nvec && (n_ranges & 7)) {                    // n_ranges % 8 keys of the last vector, one by one
                for (int e = 0; e < (n_ranges & 7); ++e) v[e >> 1] |= (uint32_t)rmax[8 * nvec + e] << (16 * (e & 1));
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) w[4 * q + x] = v[x];            // (already keys; padding: key 0)
        }
    }
    {
        u32x4* hz = reinterpret_cast<u32x4*>(s_hist);
#pragma unroll
        for (int q = 0; q < T2_BINS * T2_COPIES / 4 / T2_THREADS; ++q) hz[tid + q * T2_THREADS] = (u32x4){0u, 0u, 0u, 0u};
    }
#ifdef SKV_TOPK_STAMPS
    if (w[0] == 0x12345678u) s_out[10] = 1;
    TOPK_STAMP(20);
#endif
    int thr_l, ne_l;
    t2_find_threshold<NW>(w, T2_THREADS * NW * 2 - n_ranges, k, tid, s_hist, s_w, s_out, thr_l, ne_l, [] {});
    TOPK_STAMP(21);
    uint32_t mc = 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int lo = (int)(w[i] & 0xffffu), hi = (int)(w[i] >> 16);
        mc |= ((uint32_t)(lo >= thr_l && r0 + 2 * i < n_ranges) | ((uint32_t)(hi >= thr_l && r0 + 2 * i + 1 < n_ranges) << 1)) << (2 * i);
    }
    const int cc = __builtin_popcount(mc);
    const int cincl = block_scan_incl1(cc, s_w + 32, tid);
    if (tid == T2_THREADS - 1) s_out[8] = cincl;
    {   // the histogram is searched again below
        u32x4* hz = reinterpret_cast<u32x4*>(s_hist);
#pragma unroll
        for (int q = 0; q < T2_BINS * T2_COPIES / 4 / T2_THREADS; ++q) hz[tid + q * T2_THREADS] = (u32x4){0u, 0u, 0u, 0u};
    }
    if (cincl <= SMP_CAND / SMP_RANGE) {
        int o = cincl - cc;
        uint32_t m = mc;
        while (m) {
            s_cidx[o++] = r0 + __builtin_ctz(m);                        // qualifying ranges, ascending
            m &= m - 1;
        }
    }
    __syncthreads();
    TOPK_STAMP(22);
    const int CR = s_out[8];
    if (CR > SMP_CAND / SMP_RANGE) return -1;
    const int n = sample_candidates_topk(row, CR * SMP_RANGE, k, tid, s_hist, s_w, s_out, s_cur,
                                         [&](int c) { return s_cidx[c / SMP_RANGE] * SMP_RANGE + (c % SMP_RANGE); });
    __syncthreads();                                                    // the winners in s_cur are visible
    return n;
}

// Exact top-k of ONE part of a logit row (Vp <= 131,072 logits starting at `row`), all 1,024 threads: the part lives in
// registers, the k-th largest of the 1,024 per-thread maxima bounds the candidates, the exact search (4,096-bin histogram
// counted down from the maximum, skv_select_front.h) runs on those few dozen.  Leaves in s_cur[0 .. n) the LOCAL ids,
// ascending, of every logit > thr and of the logits == thr (thr = the k-th largest value) in ascending id order - ALL of
// them, as the reference's filter keeps every logit that is not < the k-th value (models/tensor_op.py:253-255), up to
// SMP_KEEP winners in all; returns n (uniform).  Ends with a barrier.
template <int SEGV>
__device__ __forceinline__ int sample_part_topk(const bf16_t* __restrict__ row, const int Vp, const int k, const int tid,
                                                int* s_hist, int* s_w, int* s_out, int* s_cidx, int* s_cur) {
    constexpr int NW = 4 * SEGV, NG = (NW + 15) / 16;
    uint32_t w[NW];
    const int j0 = tid * SEGV * 8;
    TOPK_STAMP(19);
    {
        const u32x4* gvec = reinterpret_cast<const u32x4*>(row);
        const int nvec = Vp / 8;
#pragma unroll
        for (int q = 0; q < SEGV; ++q) {
            const int vi = tid * SEGV + q;
            const u32x4 v = vi < nvec ? gvec[vi] : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int x = 0; x < 4; ++x) w[4 * q + x] = vi < nvec ? bf16x2_to_keys(v[x]) : 0u;   // padding: key 0
        }
    }
    {
        u32x4* hz = reinterpret_cast<u32x4*>(s_hist);
#pragma unroll
        for (int q = 0; q < T2_BINS * T2_COPIES / 4 / T2_THREADS; ++q) hz[tid + q * T2_THREADS] = (u32x4){0u, 0u, 0u, 0u};
    }
    // ---- prefilter: the k-th largest of the 1,024 per-thread maxima is a lower bound L of the k-th largest logit (k
    // distinct logits are >= L), so only logits >= L can be winners: a few dozen of 128 K.  They are compacted (token
    // order) and the exact search runs on them: one histogram atomic per thread instead of 8 * SEGV * ... per thread.
    bool done = false;
    int n_out = 0;
    {
        uint32_t m2 = w[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) m2 = pk_max_u16(m2, w[i]);
        const uint32_t w1[1] = {max(m2 & 0xffffu, m2 >> 16)};               // (high half: padding key 0)
#ifdef SKV_TOPK_STAMPS
        if (w1[0] == 0x12345678u) s_out[10] = 1;     // waits for the row
        TOPK_STAMP(20);
#endif
        int thr_l, ne_l;
        t2_find_threshold<1>(w1, T2_THREADS, k, tid, s_hist, s_w, s_out, thr_l, ne_l, [] {});
        TOPK_STAMP(21);
        uint32_t mc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) mc[g] = 0u;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int lo = (int)(w[i] & 0xffffu), hi = (int)(w[i] >> 16);
            mc[i / 16] |= ((uint32_t)(lo >= thr_l) | ((uint32_t)(hi >= thr_l) << 1)) << (2 * (i % 16));
        }
        int cc = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) cc += __builtin_popcount(mc[g]);
        const int cincl = block_scan_incl1(cc, s_w + 32, tid);
        if (tid == T2_THREADS - 1) s_out[8] = cincl;
        {   // the histogram is searched again below (either path)
            u32x4* hz = reinterpret_cast<u32x4*>(s_hist);
#pragma unroll
            for (int q = 0; q < T2_BINS * T2_COPIES / 4 / T2_THREADS; ++q) hz[tid + q * T2_THREADS] = (u32x4){0u, 0u, 0u, 0u};
        }
        int o = cincl - cc;
        if (cincl <= SMP_CAND) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint32_t m = mc[g];
                while (m) {
                    s_cidx[o++] = j0 + g * 32 + __builtin_ctz(m);
                    m &= m - 1;
                }
            }
        }
        __syncthreads();
        TOPK_STAMP(22);
        const int C = s_out[8];
        if (C <= SMP_CAND) {                                              // (always, unless thousands of logits tie)
            n_out = sample_candidates_topk(row, C, k, tid, s_hist, s_w, s_out, s_cur, [&](int c) { return s_cidx[c]; });
            done = true;
        }
    }
    if (!done) {
        int thr, need_eq;
        t2_find_threshold<NW>(w, T2_THREADS * SEGV * 8 - Vp, k, tid, s_hist, s_w, s_out, thr, need_eq, [] {});
        // flags (keys use all 16 bits here: plain compares), ordered compaction as in skv_topk2_kernel
        uint32_t mge[NG], mgt[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            mge[g] = 0u;
            mgt[g] = 0u;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int lo = (int)(w[i] & 0xffffu), hi = (int)(w[i] >> 16);
            const uint32_t fe = (uint32_t)(lo >= thr) | ((uint32_t)(hi >= thr) << 1), fg = (uint32_t)(lo > thr) | ((uint32_t)(hi > thr) << 1);
            mge[i / 16] |= fe << (2 * (i % 16));
            mgt[i / 16] |= fg << (2 * (i % 16));
        }
        if (thr == 0) {   // (padding has key 0 and the highest ids; the k-th value can only be 0 when every logit of the part is)
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int j = j0 + 2 * i;
                const uint32_t drop = (uint32_t)(j >= Vp) | ((uint32_t)(j + 1 >= Vp) << 1);
                mge[i / 16] &= ~(drop << (2 * (i % 16)));
            }
        }
        int cg = 0, ce = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cg += __builtin_popcount(mgt[g]);
            ce += __builtin_popcount(mge[g]);
        }
        ce -= cg;
        const int packed = cg | (ce << 10);                               // (< 64 greater, <= 131,072 equal in all)
        const int pincl = block_scan_incl1(packed, s_w + 48, tid);
        if (tid == T2_THREADS - 1) s_out[9] = pincl >> 10;
        __syncthreads();
        const int n_gt = k - need_eq, keep_eq = min(s_out[9], SMP_KEEP - n_gt);
        const int pexcl = pincl - packed;
        {
            int gt_run = pexcl & 1023, eq_run = pexcl >> 10;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint32_t m = mge[g];
                while (m) {
                    const int e = __builtin_ctz(m);
                    m &= m - 1;
                    int p = -1;
                    if ((mgt[g] >> e) & 1u) {
                        p = gt_run + min(eq_run, keep_eq);
                        ++gt_run;
                    } else {
                        if (eq_run < keep_eq) p = gt_run + eq_run;
                        ++eq_run;
                    }
                    if (p >= 0) {
                        s_cur[p] = j0 + g * 32 + e;
                        s_cur[SMP_KEEP + p] = -1;                         // key not at hand: read back by the caller
                    }
                }
            }
        }
        n_out = n_gt + keep_eq;
    }
    __syncthreads();
    return n_out;
}

template <int SEGV>
__global__ __launch_bounds__(T2_THREADS) void skv_sample_topk_kernel(
    const bf16_t* __restrict__ logits, long long row_stride, int V, int parts, int part_len, int k, float temperature,
    float top_p, unsigned long long seed, int64_t* __restrict__ token, int64_t* __restrict__ pos, int64_t* __restrict__ gen,
    int64_t* __restrict__ row_idx, int32_t* __restrict__ kv_len, int64_t* __restrict__ step_idx, long long base,
    long long slack, long long table_len, const int32_t* __restrict__ hit_cnts, int n_hit_cnts,
    int64_t* __restrict__ hit_accum, const uint16_t* __restrict__ range_max /* nullable: [bs][range_stride] */,
    long long range_stride) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    int* s_hist = smem;                               // [T2_BINS][T2_COPIES]
    int* s_w = s_hist + T2_BINS * T2_COPIES;          // [80]
    int* s_out = s_w + 80;                            // [16]
    int* s_cur = s_out + 16;                          // [2][64] local token id per winner of the current part (ascending id), its key
    int* s_cidx = s_cur + 2 * SMP_KEEP;               // [SMP_CAND] token ids of the prefilter's candidates, ascending
    float* s_sv = reinterpret_cast<float*>(s_cidx + SMP_CAND);   // [64] sorted: logit / temperature, descending
    int* s_si = reinterpret_cast<int*>(s_sv + 64);    // [64] sorted: token id
    // [SMP_PARTS * 64] (key + 1) << 32 | (0x7fffffff - token id) of every part's winners (16-byte aligned: see the launcher)
    unsigned long long* s_cc = reinterpret_cast<unsigned long long*>(s_si + 64);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bf16_t* const row = logits + (size_t)b * row_stride;
    // ---- through the lm_head's range maxima when it left them: the row is not streamed at all
    if (range_max != nullptr) {
        const int n = sample_ranges_topk(row, range_max + (size_t)b * range_stride, V / SMP_RANGE, k, tid, s_hist, s_w, s_out,
                                         s_cidx, s_cur);
        if (n >= 0) {                                        // (uniform)
            if (tid < SMP_KEEP) {
                const int id = tid < n ? s_cur[tid] : 0x7fffffff;
                const unsigned long long kv = tid < n ? (unsigned long long)((uint32_t)s_cur[SMP_KEEP + tid] + 1u) : 0ull;
                s_cc[tid] = (kv << 32) | (unsigned long long)(uint32_t)(0x7fffffff - id);
            }
            if (tid == 0) s_out[12] = 0;
            __syncthreads();
            TOPK_STAMP(27);
            parts = 0;                                       // no part is searched; the merge ranks the SMP_KEEP entries above
        }
    }
    // ---- the winners of every part (a row of <= 131,072 logits is one part)
    for (int p = 0; p < parts; ++p) {
        const int v0 = p * part_len, vp = min(part_len, V - v0);
        const int n = sample_part_topk<SEGV>(row + v0, vp, min(k, vp), tid, s_hist, s_w, s_out, s_cidx, s_cur);
        if (tid < SMP_KEEP) {
            const int id = tid < n ? v0 + s_cur[tid] : 0x7fffffff;
            int wk = tid < n ? s_cur[SMP_KEEP + tid] : 0;
            if (wk < 0) wk = (int)(bf16x2_to_keys((uint32_t)row[id]) & 0xffffu);   // (the all-slots path: logit read back, L2-hot)
            // (key + 1, ~id) as ONE 64-bit number: its order is "value descending, token id ascending"; 0 = no winner
            const unsigned long long kv = tid < n ? (unsigned long long)((uint32_t)wk + 1u) : 0ull;
            s_cc[p * SMP_KEEP + tid] = (kv << 32) | (unsigned long long)(uint32_t)(0x7fffffff - id);
        }
        if (tid == 0 && p == 0) s_out[12] = 0;
        __syncthreads();
        TOPK_STAMP(27);
    }
    // ---- merge: rank every winner by (value descending, token id ascending); the k-th of them is the k-th largest logit
    // of the row; everything not below it stays (up to SMP_KEEP), already in sorted order by its rank
    const int nc = max(parts, 1) * SMP_KEEP;
    int key = -1, id = 0x7fffffff, rank = 0x7fffffff;
    if (tid < nc) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const unsigned long long mine = s_cc[tid];
        key = (int)(mine >> 32) - 1;
        id = 0x7fffffff - (int)(uint32_t)mine;
        rank = 0;                                     // winners ahead of this one (the composites of real winners are distinct)
#pragma unroll 8
        for (int m = 0; m < nc; m += 2) {             // (one 16-B LDS read per two candidates, all lanes the same address)
            const u64x2 o = *reinterpret_cast<const u64x2*>(s_cc + m);
            rank += (o.x > mine) + (o.y > mine);
        }
        if (key >= 0 && rank == k - 1) s_out[11] = key;
    }
    __syncthreads();
    if (tid < nc && key >= s_out[11]) atomicMax(&s_out[12], rank + 1);     // (key >= thr >= 0: a winner)
    __syncthreads();
    const int kk = min(s_out[12], SMP_KEEP);
    TOPK_STAMP(28);
    if (tid < nc && rank < kk) {
        s_sv[rank] = key_to_float(key) / temperature;
        s_si[rank] = id;
    }
    __syncthreads();
    if (tid >= 64) return;
    // ---- top-p, draw, counters: as skv_sample_advance_kernel
    const long long p0 = pos[b];
    const float v = lane < kk ? s_sv[lane] : -INFINITY;
    const float mx = wave_max_dpp(v);
    const float e = lane < kk ? __expf(v - mx) : 0.f;
    const float tot = wave_tree_sum(e);
    const float p = e / tot;
    float c = p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(c, o, 64);
        if (lane >= o) c += n;
    }
    const bool keep = lane < kk && (top_p <= 0.f || lane == 0 || (c - p) <= top_p);
    const float pk = keep ? e : 0.f;
    uint32_t h = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9e3779b9u * (uint32_t)p0));
    h = mix32(h ^ (0x85ebca6bu * (uint32_t)(b + 1)) ^ (0xc2b2ae35u * (uint32_t)(lane + 1)) ^ (uint32_t)(p0 >> 32));
    const float u = ((h >> 8) + 1) * (1.0f / 16777216.0f);
    const float ex = -__logf(u);
    const float score = keep ? pk / fmaxf(ex, 1e-30f) : -1.f;
    const float best = wave_max_dpp(score);
    const unsigned long long mwin = __ballot(score == best);
    const int win = __ffsll((long long)mwin) - 1;
    if (lane == win) token[b] = (int64_t)s_si[lane];
    if (b == 0 && hit_accum != nullptr) {                 // statistics: chunk hits of this step (all layers' cnts)
        int hsum = 0;
        for (int i = lane; i < n_hit_cnts; i += 64) hsum += hit_cnts[i];
        hsum = wave_sum_i32(hsum);
        if (lane == 0) hit_accum[0] += hsum;
    }
    TOPK_STAMP(29);
    if (lane == 0) {
        pos[b] = p0 + 1;
        if (b == 0) {
            const long long g2 = gen[0] + 1;
            gen[0] = g2;
            row_idx[0] = base + g2 % slack;
            kv_len[0] = (int32_t)(base + (g2 + 1 < slack ? g2 + 1 : slack));
            if (step_idx) step_idx[0] = (step_idx[0] + 1) % table_len;
        }
    }
}

template <int SEGV>
static int launch_sample_topk(const void* logits, long long row_stride, int V, int parts, int part_len, int bs, int k,
                              float temperature, float top_p, unsigned long long seed, int64_t* token, int64_t* pos,
                              int64_t* gen, int64_t* row_idx, int32_t* kv_len, int64_t* step_idx, long long base,
                              long long slack, long long table_len, const int32_t* hit_cnts, int n_hit_cnts,
                              int64_t* hit_accum, const uint16_t* range_max, long long range_stride, hipStream_t st) {
    const size_t smem = (size_t)(T2_BINS * T2_COPIES + 80 + 16 + 64 * 4 + SMP_CAND + 2 * SMP_PARTS * SMP_KEEP) * sizeof(int);
    static size_t attr_bytes[64] = {};
    if (skv_ensure_max_lds((const void*)skv_sample_topk_kernel<SEGV>, smem, attr_bytes) != SKV_OK) return SKV_ERR_LAUNCH;
    hipLaunchKernelGGL(skv_sample_topk_kernel<SEGV>, dim3(bs), dim3(T2_THREADS), smem, st, (const bf16_t*)logits, row_stride,
                       V, parts, part_len, k, temperature, top_p, seed, token, pos, gen, row_idx, kv_len, step_idx, base, slack,
                       table_len, hit_cnts, n_hit_cnts, hit_accum, range_max, range_stride);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

static int sample_topk_advance(const void* logits, long long row_stride, int vocab, int batch_size, int k,
                              float temperature, float top_p, unsigned long long seed, int64_t* token, int64_t* pos,
                              int64_t* gen, int64_t* row_idx, int32_t* kv_len, int64_t* step_idx, long long base,
                              long long slack, long long table_len, const int32_t* hit_cnts, int n_hit_cnts,
                              int64_t* hit_accum, const uint16_t* range_max, long long range_stride, skv_stream_t stream) {
    if (!logits || !token || !pos || !gen || !row_idx || !kv_len || batch_size < 1 || !(temperature > 0.f)) return SKV_ERR_ARG;
    if (k < 1 || k > 64 || vocab < k || slack < 1 || (step_idx && table_len < 1)) return SKV_ERR_UNSUPPORTED;
    constexpr int PART_MAX = T2_THREADS * 16 * 8;          // logits one pass holds in registers
    if ((vocab % 8) || (row_stride % 8) || (((size_t)logits) & 15) || vocab > SMP_PARTS * PART_MAX) return SKV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // rows beyond 131,072 logits (GLM-4: 151,552) are searched in equal parts, each a multiple of 8 logits
    const int parts = (vocab + PART_MAX - 1) / PART_MAX;
    const int part_len = (((vocab + parts - 1) / parts) + 7) & ~7;
    if (vocab - (parts - 1) * part_len < k) return SKV_ERR_UNSUPPORTED;   // (never for real vocabularies)
    const int per_thread = (part_len / 8 + T2_THREADS - 1) / T2_THREADS;
    if (range_max) {    // 16-logit ranges, 16 keys per thread, 16-B vectors of keys
        if ((vocab % SMP_RANGE) || vocab / SMP_RANGE > T2_THREADS * 16 || vocab / SMP_RANGE < k || (((size_t)range_max) & 15) ||
            (range_stride % 8) || range_stride < vocab / SMP_RANGE)
            return SKV_ERR_UNSUPPORTED;
    }
#define SKV_ST(SV) launch_sample_topk<SV>(logits, row_stride, vocab, parts, part_len, batch_size, k, temperature, top_p, seed, \
                                          token, pos, gen, row_idx, kv_len, step_idx, base, slack, table_len, hit_cnts,         \
                                          n_hit_cnts, hit_accum, range_max, range_stride, st)
    if (per_thread <= 1) return SKV_ST(1);
    if (per_thread <= 2) return SKV_ST(2);
    if (per_thread <= 4) return SKV_ST(4);
    if (per_thread <= 8) return SKV_ST(8);
    return SKV_ST(16);
#undef SKV_ST
}

extern "C" int skv_sample_topk_advance(const void* logits, long long row_stride, int vocab, int batch_size, int k,
                                       float temperature, float top_p, unsigned long long seed, int64_t* token, int64_t* pos,
                                       int64_t* gen, int64_t* row_idx, int32_t* kv_len, int64_t* step_idx, long long base,
                                       long long slack, long long table_len, const int32_t* hit_cnts, int n_hit_cnts,
                                       int64_t* hit_accum, skv_stream_t stream) {
    return sample_topk_advance(logits, row_stride, vocab, batch_size, k, temperature, top_p, seed, token, pos, gen, row_idx, kv_len,
                               step_idx, base, slack, table_len, hit_cnts, n_hit_cnts, hit_accum, nullptr, 0, stream);
}

extern "C" int skv_sample_topk_advance_ranges(const void* logits, long long row_stride, int vocab, const void* range_max,
                                              long long range_stride, int batch_size, int k, float temperature, float top_p,
                                              unsigned long long seed, int64_t* token, int64_t* pos, int64_t* gen,
                                              int64_t* row_idx, int32_t* kv_len, int64_t* step_idx, long long base,
                                              long long slack, long long table_len, const int32_t* hit_cnts, int n_hit_cnts,
                                              int64_t* hit_accum, skv_stream_t stream) {
    if (!range_max) return SKV_ERR_ARG;
    return sample_topk_advance(logits, row_stride, vocab, batch_size, k, temperature, top_p, seed, token, pos, gen, row_idx, kv_len,
                               step_idx, base, slack, table_len, hit_cnts, n_hit_cnts, hit_accum, (const uint16_t*)range_max,
                               range_stride, stream);
}

extern "C" int skv_sample_advance(const float* vals, const int64_t* idx, int batch_size, int k, float top_p,
                                  unsigned long long seed, int64_t* token, int64_t* pos, int64_t* gen, int64_t* row_idx,
                                  int32_t* kv_len, int64_t* step_idx, long long base, long long slack,
                                  long long table_len, const int32_t* hit_cnts, int n_hit_cnts, int64_t* hit_accum,
                                  skv_stream_t stream) {
    if (!vals || !idx || !token || !pos || !gen || !row_idx || !kv_len || batch_size < 1) return SKV_ERR_ARG;
    if (k < 1 || k > 64 || slack < 1 || (step_idx && table_len < 1)) return SKV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(skv_sample_advance_kernel, dim3(batch_size), dim3(64), 0, (hipStream_t)stream, vals, idx, k, top_p,
                       seed, token, pos, gen, row_idx, kv_len, step_idx, base, slack, table_len, hit_cnts, n_hit_cnts, hit_accum);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

#ifdef SKV_TOPK_STAMPS
// diagnostic builds only (libshadowkv_hip_stamps.so, tools/sample_stamps.py): the phase stamps of the last sampler launch
extern "C" __attribute__((visibility("default"))) int skv_debug_sample_stamps(unsigned long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_topk_stamps), sizeof(g_topk_stamps)) == hipSuccess ? 0 : -1;
}
#endif
