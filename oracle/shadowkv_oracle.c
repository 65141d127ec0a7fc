/*
 * shadowkv_oracle.c -- CPU restatement of the ShadowKV decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under shadowkv_amd/ may import, link or
 * call this file.  Only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py use it, as the checker / the reported CPU baseline.
 *
 * Every function restates one stage of the reference's decode path with the
 * rounding points of the reference's *kernels* (SURVEY.md section 8a, table
 * "Rounding points").  Citations are into /root/reference.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - reorder / gather / d2d compaction: integer- and byte-exact, pinned by the
 *     golden models of kernels/test_cached_gather_copy.cu:70-216 and by traces
 *     of the reference's pure-PyTorch ShadowKVCache (tests/golden/).
 *   - K rebuild + RoPE, V gather, landmark build: pinned by fixtures generated
 *     from models/kv_cache.py:155-506 (tests/golden/make_golden.py).
 *   - softmax statistics (CUTLASS EpilogueVisitorSoftmax, un-vendored) and
 *     torch.topk tie order: PARITY UNPINNED against the CUDA path; the contract
 *     used here is written out below and in DESIGN.md.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp)
 */
#define _GNU_SOURCE
#include <math.h>
#include <omp.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* bf16 helpers                                                              */
/* ------------------------------------------------------------------------- */
static inline float bf2f(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* round-to-nearest-even f32 -> bf16, NaN stays NaN (what v_cvt_pk_bf16_f32 and
 * CUDA __float2bfloat16_rn do). */
static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

/* bf16 arithmetic with one rounding per op (CUDA __hmul / __hadd on
 * __nv_bfloat16: kernels/rope_new.cu:366-367). */
static inline uint16_t bf_mul(uint16_t a, uint16_t b) { return f2bf(bf2f(a) * bf2f(b)); }
static inline uint16_t bf_add(uint16_t a, uint16_t b) { return f2bf(bf2f(a) + bf2f(b)); }
static inline uint16_t bf_neg(uint16_t a) { return (uint16_t)(a ^ 0x8000u); }

/* ------------------------------------------------------------------------- */
/* a4: landmark scoring  (models/kv_cache.py:1006-1019 -> batch_gemm_softmax) */
/* ------------------------------------------------------------------------- */

/* Contract of the f32 dot product (k = 128 = 16 lanes x 8 contiguous elements):
 * lane l accumulates its 8 products as a sequential fma chain starting from 0,
 * the 16 lane partials are summed by a balanced binary tree in natural order
 * ((p0+p1)+(p2+p3))+...  bf16 x bf16 products are exact in f32, so fma == add
 * of the exact product and the result is fully determined by this order.
 * General k (multiple of 8): lanes = k/8 must be a power of two <= 64, the tree
 * is the same balanced tree over `lanes` partials. */
static float score_dot(const uint16_t *q, const uint16_t *x, int k) {
    float p[64];
    int lanes = k / 8;
    for (int l = 0; l < lanes; ++l) {
        float acc = 0.0f;
        for (int j = 0; j < 8; ++j) acc = fmaf(bf2f(q[8 * l + j]), bf2f(x[8 * l + j]), acc);
        p[l] = acc;
    }
    for (int w = lanes; w > 1; w >>= 1)
        for (int i = 0; i < w / 2; ++i) p[i] = p[2 * i] + p[2 * i + 1];
    return p[0];
}

/* exp for x <= 0 with a fully specified f32 op sequence (the reference uses
 * CUTLASS fast_exp = ex2.approx, batch_gemm_softmax.h:248; not reproducible
 * off-GPU, so the contract is this one; max rel. error ~1.5e-7):
 *   x < -80 -> 0;  t = x*log2e; n = rint(t); r = fma(n,-ln2_hi,x); r = fma(n,-ln2_lo,r)
 *   p = Horner degree 6 of exp(r) with fma; result = p * 2^n by exponent add. */
static inline float spec_exp(float x) {
    if (!(x >= -80.0f)) return 0.0f;
    const float LOG2E = 1.44269504088896341f;
    const float LN2_HI = 0.693145751953125f;      /* 0x3f317200 */
    const float LN2_LO = 1.42860682030941723e-6f; /* ln2 - LN2_HI */
    float t = x * LOG2E;
    float n = rintf(t);
    float r = fmaf(n, -LN2_HI, x);
    r = fmaf(n, -LN2_LO, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int32_t bits;
    memcpy(&bits, &p, 4);
    bits += ((int32_t)n) << 23;
    memcpy(&p, &bits, 4);
    return p;
}

/* e in [0, 2) -> trunc(e * 2^36) as an integer; sums of these are associative,
 * so the softmax denominator does not depend on the summation order. */
static inline uint64_t exp_to_fixed(float e) {
    uint32_t bits;
    memcpy(&bits, &e, 4);
    int ex = (int)((bits >> 23) & 0xff);
    if (ex == 0) return 0; /* zero / denormal: below 2^-36 anyway */
    uint64_t mant = (uint64_t)((bits & 0x7fffffu) | 0x800000u);
    int sh = ex - 127 - 23 + 36;
    if (sh >= 0) return mant << sh;
    if (sh <= -24) return 0;
    return mant >> (-sh);
}

/* batch_gemm_softmax (kernels/functions.h:460, kernels/batch_gemm_softmax.cu:229-293,
 * kernels/batch_gemm_softmax.h:207-302, :523-614).
 *   A [batch][m][k] bf16 row-major (q), B [batch][n][k] bf16 (landmarks),
 *   D, Softmax [batch][m][n] bf16, Norm / Sum [batch][T][m] f32, T = ceil(n/256)
 *   (ldn = lds = m, batch stride T*m: batch_gemm_softmax.cu:249-271).
 * Same three stages as the reference (GEMM + per-256-column-tile partials, final
 * reduction, apply), with every f32 operation order written out:
 *  1. D_j   = bf16(alpha * dot_j)                       f32 mul, one bf16 rounding
 *     per tile t (columns 256t..256t+255):
 *       m_t = max_j float(D_j)                           exact
 *       S_t = sum_j trunc(2^36 * exp(float(D_j) - m_t))  integer, order-free
 *       s_t = (float)(S_t * 2^-36)                       one rounding
 *     Norm[b][t][row] = m_t, Sum[b][t][row] = s_t        (CUTLASS EpilogueVisitorSoftmax
 *     takes max/sum over the bf16-rounded D; un-vendored, see header)
 *  2. m = max_t m_t;  w_t = s_t * exp(m_t - m);
 *     s = lane-strided sum: lane l (0..63) adds w_l, w_{l+64}, ... in order starting
 *     from 0, then a balanced binary tree over the 64 lanes in natural order;
 *     inv = 1/s;  Norm[b][0][row] <- m, Sum[b][0][row] <- inv (tile-0 slots are what
 *     the apply stage reads back, batch_gemm_softmax.h:274-275)
 *  3. P_j = bf16(exp(float(D_j) - m) * inv)              batch_gemm_softmax.h:289-292
 * beta is ignored (ScaleType::OnlyAlphaScaling, batch_gemm_softmax.cu:157-163). */
static void softmax_finalize(const float *mt, const float *st, int T, int stride, float *m_out,
                             float *inv_out) {
    float m = -INFINITY;
    for (int t = 0; t < T; ++t)
        if (mt[(size_t)t * stride] > m) m = mt[(size_t)t * stride];
    float lane[64];
    for (int l = 0; l < 64; ++l) {
        float acc = 0.0f;
        for (int t = l; t < T; t += 64)
            acc = acc + st[(size_t)t * stride] * spec_exp(mt[(size_t)t * stride] - m);
        lane[l] = acc;
    }
    for (int w = 64; w > 1; w >>= 1)
        for (int i = 0; i < w / 2; ++i) lane[i] = lane[2 * i] + lane[2 * i + 1];
    *m_out = m;
    *inv_out = 1.0f / lane[0];
}

ORACLE_API void oracle_batch_gemm_softmax(const uint16_t *A, const uint16_t *B, uint16_t *D,
                                          float *Norm, float *Sum, uint16_t *Softmax,
                                          int batch_count, int m, int n, int k, float alpha,
                                          float beta) {
    (void)beta;
    int T = (n + 255) / 256;
    /* three phases like the reference's three launches; every element is computed exactly as in the serial form, the
     * loops are only spread over (batch, row, tile) so that all host cores take part (cpu_baseline in bench.py) */
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < batch_count; ++b) {
        for (int r = 0; r < m; ++r) {
            for (int t = 0; t < T; ++t) {
                const uint16_t *q = A + ((size_t)b * m + r) * k;
                const uint16_t *Bb = B + (size_t)b * n * k;
                uint16_t *Dr = D + ((size_t)b * m + r) * n;
                float *mt = Norm + (size_t)b * T * m + r; /* [t*m] */
                float *st = Sum + (size_t)b * T * m + r;
                int j0 = t * 256, j1 = j0 + 256 < n ? j0 + 256 : n;
                float mx = -INFINITY;
                for (int j = j0; j < j1; ++j) {
                    uint16_t d = f2bf(alpha * score_dot(q, Bb + (size_t)j * k, k));
                    Dr[j] = d;
                    if (bf2f(d) > mx) mx = bf2f(d);
                }
                uint64_t S = 0;
                for (int j = j0; j < j1; ++j) S += exp_to_fixed(spec_exp(bf2f(Dr[j]) - mx));
                mt[(size_t)t * m] = mx;
                st[(size_t)t * m] = (float)((double)S * (1.0 / 68719476736.0));
            }
        }
    }
    float *fin = (float *)malloc((size_t)batch_count * m * 2 * sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch_count; ++b)
        for (int r = 0; r < m; ++r)
            softmax_finalize(Norm + (size_t)b * T * m + r, Sum + (size_t)b * T * m + r, T, m,
                             &fin[((size_t)b * m + r) * 2], &fin[((size_t)b * m + r) * 2 + 1]);
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < batch_count; ++b)
        for (int r = 0; r < m; ++r)
            for (int t = 0; t < T; ++t) {
                const uint16_t *Dr = D + ((size_t)b * m + r) * n;
                uint16_t *Pr = Softmax + ((size_t)b * m + r) * n;
                const float mfin = fin[((size_t)b * m + r) * 2], inv = fin[((size_t)b * m + r) * 2 + 1];
                int j0 = t * 256, j1 = j0 + 256 < n ? j0 + 256 : n;
                for (int j = j0; j < j1; ++j) Pr[j] = f2bf(spec_exp(bf2f(Dr[j]) - mfin) * inv);
            }
    for (int b = 0; b < batch_count; ++b)
        for (int r = 0; r < m; ++r) {
            Norm[(size_t)b * T * m + r] = fin[((size_t)b * m + r) * 2];
            Sum[(size_t)b * T * m + r] = fin[((size_t)b * m + r) * 2 + 1];
        }
    free(fin);
}

/* ------------------------------------------------------------------------- */
/* a5: group max + top-k + slot->chunk map (models/kv_cache.py:1023-1042)    */
/* ------------------------------------------------------------------------- */
/* P [blocks][groups][n] bf16 (all >= 0).  score[j] = max_g P[g][j] (exact, bf16).
 * Selects the `topk` largest scores; membership under ties is NOT defined by
 * the reference (torch.topk, CUDA) -- contract here: ties at the threshold go to
 * the LOWEST landmark slot.  Output order: ascending landmark slot.  If
 * landmark_idx != NULL the slot is mapped to its chunk id (k_landmark_idx
 * gather, kv_cache.py:1039-1042).  Returns 0, or -1 if n < topk. */
static int cmp_bf16_desc(const void *a, const void *c) {
    float fa = bf2f(*(const uint16_t *)a), fc = bf2f(*(const uint16_t *)c);
    return (fa < fc) - (fa > fc);
}

ORACLE_API int oracle_group_max_topk(const uint16_t *P, const int64_t *landmark_idx, int64_t *out,
                                     int blocks, int groups, int n, int topk) {
    if (n < topk) return -1;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < blocks; ++b) {
        uint16_t *score = (uint16_t *)malloc((size_t)n * sizeof(uint16_t));
        uint16_t *tmp = (uint16_t *)malloc((size_t)n * sizeof(uint16_t));
        const uint16_t *Pb = P + (size_t)b * groups * n;
        for (int j = 0; j < n; ++j) {
            uint16_t mx = Pb[j];
            for (int g = 1; g < groups; ++g) {
                uint16_t v = Pb[(size_t)g * n + j];
                if (bf2f(v) > bf2f(mx)) mx = v;
            }
            score[j] = mx;
        }
        /* threshold = topk-th largest value */
        memcpy(tmp, score, (size_t)n * sizeof(uint16_t));
        qsort(tmp, (size_t)n, sizeof(uint16_t), cmp_bf16_desc);
        float fthr = bf2f(tmp[topk - 1]);
        int n_gt = 0;
        for (int j = 0; j < n; ++j) n_gt += bf2f(score[j]) > fthr;
        int need_eq = topk - n_gt;
        int w = 0;
        for (int j = 0; j < n && w < topk; ++j) {
            float v = bf2f(score[j]);
            int take = 0;
            if (v > fthr) take = 1;
            else if (v == fthr && need_eq > 0) { take = 1; --need_eq; }
            if (take) {
                out[(size_t)b * topk + w] = landmark_idx ? landmark_idx[(size_t)b * n + j] : (int64_t)j;
                ++w;
            }
        }
        free(score);
        free(tmp);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a6: chunk-cache diff (kernels/map.cuh:754-796, gather_copy.cu:259-307)    */
/* ------------------------------------------------------------------------- */
/* Per block (bs*heads of them): `cached` = the map_size chunk ids resident in
 * the sparse region (slot i holds chunk cached[i]); `cur` = the map_size ids
 * selected this step.  Outputs (cached is overwritten in place):
 *   hits   = cur ids present in cached, sorted by their OLD slot (block_sort2
 *            pass 1, map.cuh:508-539); offsets = old slot
 *   misses = the rest, sorted ascending by chunk id (pass 2); offsets = chunk id
 *   cnts[b] = number of hits
 * Duplicate ids in `cached`: the reference's hash insert is racy; here the
 * LOWEST slot wins, as in the golden model (std::map::insert keeps the first,
 * test_cached_gather_copy.cu:83-87).  Ids are truncated to int32 like
 * map.cuh:771,776.  Negative ids (the -1 "empty" sentinel position_ids start with,
 * kv_cache.py:634) never match.  Unlike the reference (silent no-op,
 * gather_copy.cu:278-306) any map_size >= 1 is accepted. */
typedef struct { int32_t off; int32_t key; } ok_pair;
static int cmp_pair_off(const void *a, const void *b) {
    const ok_pair *x = (const ok_pair *)a, *y = (const ok_pair *)b;
    if (x->off != y->off) return x->off < y->off ? -1 : 1;
    return 0;
}
static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

ORACLE_API void oracle_reorder_keys_and_compute_offsets(int64_t *cached_pos_ids,
                                                        const int64_t *cur_pos_ids,
                                                        int32_t *offsets, int32_t *cnts,
                                                        int batch_size, int heads, int map_size) {
    int blocks = batch_size * heads;
    ok_pair *hits = (ok_pair *)malloc((size_t)map_size * sizeof(ok_pair));
    int32_t *miss = (int32_t *)malloc((size_t)map_size * sizeof(int32_t));
    for (int b = 0; b < blocks; ++b) {
        int64_t *old = cached_pos_ids + (size_t)b * map_size;
        const int64_t *cur = cur_pos_ids + (size_t)b * map_size;
        int nh = 0, nm = 0;
        for (int i = 0; i < map_size; ++i) {
            int32_t key = (int32_t)cur[i];
            int slot = -1;
            if (key >= 0)
                for (int s = 0; s < map_size; ++s)
                    if ((int32_t)old[s] == key) { slot = s; break; }
            if (slot >= 0) { hits[nh].off = slot; hits[nh].key = key; ++nh; }
            else miss[nm++] = key;
        }
        /* stable w.r.t. equal offsets is irrelevant: equal offsets imply equal keys */
        qsort(hits, (size_t)nh, sizeof(ok_pair), cmp_pair_off);
        qsort(miss, (size_t)nm, sizeof(int32_t), cmp_i32);
        int32_t *off = offsets + (size_t)b * map_size;
        for (int i = 0; i < nh; ++i) { old[i] = hits[i].key; off[i] = hits[i].off; }
        for (int i = 0; i < nm; ++i) { old[nh + i] = miss[i]; off[nh + i] = miss[i]; }
        cnts[b] = nh;
    }
    free(hits);
    free(miss);
}

/* ------------------------------------------------------------------------- */
/* a7 / a8: chunk-row movement (kernels/copy.cuh:303-362, :649-687, :785-846) */
/* ------------------------------------------------------------------------- */
#define ROW_ELEMS 1024 /* one chunk row = 128 lanes x 16 B = 1024 bf16 (copy.cuh BLOCK_SIZE_CP x int4) */

/* In-place compaction of hit rows inside the sparse region of a [blocks][stride]
 * bf16 buffer: row i <- row offsets[i], i < cnt, with snapshot semantics (every
 * source row is read before any row it could alias is written; the reference
 * gets this from offsets[i] >= i and 32-row staging). Lengths/offsets are in
 * bf16 elements like the reference's arguments (kv_cache.py:1090-1093). */
static void compact_rows(uint16_t *buf, const int32_t *offsets, int cnt, size_t base) {
    uint16_t *tmp = (uint16_t *)malloc((size_t)(cnt > 0 ? cnt : 1) * ROW_ELEMS * 2);
    for (int i = 0; i < cnt; ++i)
        memcpy(tmp + (size_t)i * ROW_ELEMS, buf + base + (size_t)offsets[i] * ROW_ELEMS, ROW_ELEMS * 2);
    for (int i = 0; i < cnt; ++i)
        memcpy(buf + base + (size_t)i * ROW_ELEMS, tmp + (size_t)i * ROW_ELEMS, ROW_ELEMS * 2);
    free(tmp);
}

/* gather_copy_d2d_with_offsets (functions.h:97, gather_copy.cu:165-240) */
ORACLE_API void oracle_gather_copy_d2d_with_offsets(uint16_t *keys, const int32_t *offsets,
                                                    const int32_t *cnts, int batch_size, int heads,
                                                    int gpu_k_length, int gpu_k_offset,
                                                    int gpu_k_stride, int map_size) {
    (void)gpu_k_length;
    for (int b = 0; b < batch_size * heads; ++b)
        compact_rows(keys, offsets + (size_t)b * map_size, cnts[b],
                     (size_t)b * (size_t)gpu_k_stride + (size_t)gpu_k_offset);
}

/* gather_copy_with_offsets (functions.h:151, gather_copy.cu:332-419,
 * copy.cuh:785-846): hits compacted in place, misses fetched from the host
 * table: row i <- values[b*cpu_v_length + offsets[i]*ROW], i in [cnt, map_size). */
ORACLE_API void oracle_gather_copy_with_offsets(const uint16_t *values, uint16_t *v_cache_buffer,
                                                const int32_t *offsets, const int32_t *cnts,
                                                int batch_size, int heads, int cpu_v_length,
                                                int gpu_v_length, int gpu_v_offset,
                                                int gpu_v_stride, int map_size) {
    (void)gpu_v_length;
    for (int b = 0; b < batch_size * heads; ++b) {
        const int32_t *off = offsets + (size_t)b * map_size;
        size_t base = (size_t)b * (size_t)gpu_v_stride + (size_t)gpu_v_offset;
        compact_rows(v_cache_buffer, off, cnts[b], base);
        for (int i = cnts[b]; i < map_size; ++i)
            memcpy(v_cache_buffer + base + (size_t)i * ROW_ELEMS,
                   values + (size_t)b * (size_t)cpu_v_length + (size_t)off[i] * ROW_ELEMS,
                   ROW_ELEMS * 2);
    }
}

/* gather_copy (functions.h:71, gather_copy.cu:81-146, copy.cuh:481-508): no
 * cache, int64 position ids, every row fetched from the host table. */
ORACLE_API void oracle_gather_copy(const uint16_t *values, uint16_t *v_cache_buffer,
                                   const int64_t *position_ids, int batch_size, int heads,
                                   int cpu_v_length, int gpu_v_length, int map_size) {
    for (int b = 0; b < batch_size * heads; ++b)
        for (int i = 0; i < map_size; ++i)
            memcpy(v_cache_buffer + (size_t)b * (size_t)gpu_v_length + (size_t)i * ROW_ELEMS,
                   values + (size_t)b * (size_t)cpu_v_length +
                       (size_t)position_ids[(size_t)b * map_size + i] * ROW_ELEMS,
                   ROW_ELEMS * 2);
}

/* ------------------------------------------------------------------------- */
/* a8: K rebuild  (kernels/batch_gather_gemm.cu:193-287)                     */
/* ------------------------------------------------------------------------- */
/* out[b][h][i][d] = bf16( sum_{j<rank} U[b][pos(i)][j] * SV[b][h][d][j] ),
 * pos(i) = position_ids[b][h][i / chunk] * chunk + i % chunk, f32 accumulation.
 * Accumulation order: the device kernel uses v_mfma_f32_16x16x32_bf16; probing the
 * instruction (tools/mfma_probe.hip, profiles/r01_mfma_probe.txt) shows that, per 32-wide
 * k-step, it adds the products in 4 groups of 8 consecutive k (one group per lane quarter),
 * each group summed (near-)exactly and chained into the f32 accumulator with one RNE rounding:
 * that model reproduces 89 % of results bit-for-bit on adversarial data, the rest differ in
 * the last f32 bit (internal alignment width, not restatable).  The oracle follows that
 * grouping; device-vs-oracle is therefore a tolerance comparison (<= 1 bf16 ulp on a small
 * fraction of values), not bit-exact -- see DESIGN.md.  The reference's own CUTLASS kernel
 * (mma.sync m16n8k16, 5 stages) has yet another order, so bit-exactness against the CUDA
 * path was never on offer.  Rows of whole 128-row tiles below cnt*chunk are not written
 * (early exit, gemm_universal_batch_gather_indices.h:736-738); callers must only rely on
 * rows >= cnt*chunk.  U is shared by all heads (batch/num_heads, ...indices.h:717). */
static float dot_mfma_order(const uint16_t *u, const uint16_t *sv, int rank) {
    float acc = 0.0f;
    for (int j0 = 0; j0 < rank; j0 += 8) {
        double s = (double)acc;
        int j1 = j0 + 8 < rank ? j0 + 8 : rank;
        for (int j = j0; j < j1; ++j) s += (double)bf2f(u[j]) * (double)bf2f(sv[j]);
        acc = (float)s;
    }
    return acc;
}

ORACLE_API void oracle_batch_gather_gemm(const uint16_t *U, const uint16_t *SV,
                                         const int32_t *position_ids, uint16_t *output,
                                         int batch_size, int heads, int seq_len, int embed_dim,
                                         int rank, int sparse_budget, int chunk_size,
                                         const int32_t *cnts) {
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < batch_size; ++b) {
        for (int h = 0; h < heads; ++h) {
            for (int i = 0; i < sparse_budget; ++i) {
                int cnt = cnts ? cnts[b * heads + h] : 0;
                int first = (cnt * chunk_size / 128) * 128;
                if (i < first) continue;
                const int32_t *pid = position_ids + ((size_t)b * heads + h) * (sparse_budget / chunk_size);
                long pos = (long)pid[i / chunk_size] * chunk_size + i % chunk_size;
                const uint16_t *u = U + ((size_t)b * seq_len + (size_t)pos) * rank;
                uint16_t *o = output + (((size_t)b * heads + h) * sparse_budget + i) * embed_dim;
                for (int d = 0; d < embed_dim; ++d) {
                    const uint16_t *sv = SV + (((size_t)b * heads + h) * embed_dim + d) * rank;
                    o[d] = f2bf(dot_mfma_order(u, sv, rank));
                }
            }
        }
    }
}

/* threads the parallel loops above run on (bench.py reports it with the cpu_baseline) */
ORACLE_API int oracle_num_threads(void) { return omp_get_max_threads(); }
ORACLE_API void oracle_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* cpu_baseline hygiene (bench.py): bind OpenMP thread t of the current team size to cpus[t % n_cpus] (one logical CPU per
 * physical core, chosen by the caller).  The OpenMP runtime is shared with torch and was initialised before any OMP_PROC_BIND
 * could be set for it, so the binding is done here, per thread, after the team size is chosen.  Thread 0 is the CALLING
 * thread: the caller restores its affinity afterwards (whole_set = 1 gives every thread of the team the whole list back).
 * Returns the number of threads bound. */
ORACLE_API int oracle_bind_threads(const int *cpus, int n_cpus, int whole_set) {
    int bound = 0;
    if (!cpus || n_cpus < 1) return 0;
#pragma omp parallel reduction(+ : bound)
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (whole_set)              /* undo: every thread may run on every listed CPU again */
            for (int i = 0; i < n_cpus; ++i) CPU_SET(cpus[i], &set);
        else
            CPU_SET(cpus[omp_get_thread_num() % n_cpus], &set);
        if (sched_setaffinity(0, sizeof(set), &set) == 0) bound += 1;
    }
    return bound;
}

/* the CPU every OpenMP thread of the current team runs on right now (diagnostic of the binding above) */
ORACLE_API int oracle_thread_cpus(int *out, int n_out) {
    int n = 0;
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        if (t < n_out) out[t] = sched_getcpu();
#pragma omp single
        n = omp_get_num_threads();
    }
    return n;
}

/* first touch: copy n bytes with the static partition the scoring / gather loops use over contiguous per-head state, so a page
 * of `dst` lands on the NUMA node of a thread that will read it */
ORACLE_API void oracle_parallel_copy(void *dst, const void *src, size_t n) {
    const size_t page = 4096, pages = (n + page - 1) / page;
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < pages; ++p) {
        const size_t o = p * page, len = o + page <= n ? page : n - o;
        memcpy((char *)dst + o, (const char *)src + o, len);
    }
}

/* ------------------------------------------------------------------------- */
/* a8: RoPE and push into the key cache (kernels/rope_new.cu)                */
/* ------------------------------------------------------------------------- */
/* NeoX half-split rotation in bf16 arithmetic, three roundings per output
 * (rope_new.cu:360-367; same roundings as the pure-torch oracle,
 * models/tensor_op.py:139-151).  If nvcc contracted hmul+hadd into an fma on
 * the CUDA build that would be two roundings; not verifiable offline. */
static inline void rope_llama_row(const uint16_t *x, const uint16_t *cs, uint16_t *o, int half) {
    for (int t = 0; t < half; ++t) {
        uint16_t x1 = x[t], x2 = x[t + half], c = cs[t], s = cs[t + half];
        uint16_t o1 = bf_add(bf_mul(x1, c), bf_mul(bf_neg(x2), s));
        uint16_t o2 = bf_add(bf_mul(x2, c), bf_mul(x1, s));
        o[t] = o1;
        o[t + half] = o2;
    }
}

/* GLM: interleaved pairs (2t,2t+1), t<32, cos=cs[t], sin=cs[t+32]; dims 64..127
 * copied (rope_new.cu:471-490). */
static inline void rope_glm_row(const uint16_t *x, const uint16_t *cs, uint16_t *o) {
    for (int t = 0; t < 32; ++t) {
        uint16_t x1 = x[2 * t], x2 = x[2 * t + 1], c = cs[t], s = cs[t + 32];
        uint16_t o1 = bf_add(bf_mul(x1, c), bf_mul(bf_neg(x2), s));
        uint16_t o2 = bf_add(bf_mul(x2, c), bf_mul(x1, s));
        o[2 * t] = o1;
        o[2 * t + 1] = o2;
    }
    for (int t = 64; t < 128; ++t) o[t] = x[t];
}

/* apply_rotary_pos_emb_push_cache_opt[_glm] (functions.h:362,:396;
 * rope_new.cu:321-411, :429-534).  Strides in elements.  Rows of chunks below
 * cnts[b][h] are skipped; rows at or past offset_end are skipped. */
ORACLE_API void oracle_rope_push_cache(const uint16_t *x, const uint16_t *cos_sin,
                                       const int32_t *position_ids, uint16_t *output_cache,
                                       const int32_t *cnts, int batch_size, int heads, int seq_len,
                                       int embed_dim, long stride_xb, long stride_xh, long stride_xs,
                                       long stride_cos_sin, long stride_pid_b, long stride_pid_h,
                                       long stride_pid_s, long stride_ob, long stride_oh,
                                       long stride_os, int offset_start, int offset_end,
                                       int half_dim, int chunk_size, int glm) {
    (void)embed_dim;
    for (int b = 0; b < batch_size; ++b)
        for (int h = 0; h < heads; ++h) {
            int cnt = cnts[b * heads + h];
            for (int s = 0; s < seq_len; ++s) {
                if (s / chunk_size < cnt) continue;
                if (offset_start + s >= offset_end) continue;
                long pid = position_ids[b * stride_pid_b + h * stride_pid_h + (s / chunk_size) * stride_pid_s];
                const uint16_t *cs = cos_sin + (pid * chunk_size + s % chunk_size) * stride_cos_sin;
                const uint16_t *xp = x + b * stride_xb + h * stride_xh + s * stride_xs;
                uint16_t *op = output_cache + b * stride_ob + h * stride_oh + (long)(offset_start + s) * stride_os;
                if (glm) rope_glm_row(xp, cs, op);
                else rope_llama_row(xp, cs, op, half_dim);
            }
        }
}

/* apply_rotary_pos_emb_new (functions.h:240, rope_new.cu:55-122): plain NeoX
 * RoPE with an int64 position per (b,h,s); output has x's strides. */
ORACLE_API void oracle_rope_new(const uint16_t *x, const uint16_t *cos_sin,
                                const int64_t *position_ids, uint16_t *output, int batch_size,
                                int heads, int seq_len, int embed_dim, long stride_xb,
                                long stride_xh, long stride_xs, long stride_cos_sin,
                                long stride_pid_b, long stride_pid_h, long stride_pid_s,
                                int half_dim) {
    (void)embed_dim;
    for (int b = 0; b < batch_size; ++b)
        for (int h = 0; h < heads; ++h)
            for (int s = 0; s < seq_len; ++s) {
                long pid = (long)position_ids[b * stride_pid_b + h * stride_pid_h + s * stride_pid_s];
                long xo = b * stride_xb + h * stride_xh + s * stride_xs;
                rope_llama_row(x + xo, cos_sin + pid * stride_cos_sin, output + xo, half_dim);
            }
}

/* ------------------------------------------------------------------------- */
/* a4 + a5, fused selection (round 4): the candidate rule, restated          */
/* ------------------------------------------------------------------------- */
/* The device's fused selection (csrc/skv_select.hip, t3_fused_front) drops the normalise launch: the scan leaves a 15-bit
 * monotone fixed-point key (1/256 per code) of kappa_j = min(max_g (D_gj - ctil_g), -2^-12) per slot (ctil = the previous step's log-normalisers), the
 * top-k launch computes this step's finals (m_g, 1 / s_g), takes as CANDIDATES every slot with
 *      key(kappa_j) >= min( key(theta), k15 ),   theta = low(k15) - (max_g delta_g - min_g delta_g) - 2^-4,
 * k15 = the S-th largest key, low(k) = the smallest f32 with key k, delta_g = (m_g + ln s_g) - ctil_g, evaluates the
 * candidates' scores exactly and selects the exact top S among them.  Why no other slot can be in the top S or tie with its
 * last member (the argument in full: skv_select.hip): with K_j = max_g (D_gj - c_g), c_g = m_g + ln s_g, one has
 * kappa_j - max delta <= K_j <= kappa_j - min delta (the clamp only lowers kappa of slots whose K is <= 0 anyway and keeps the
 * map monotone); S slots a have kappa_a >= low(k15); key(kappa_j) < key(theta) implies kappa_j < theta, hence
 * K_j < K_a - 2^-4 for those S slots, and score = max_g bf16(exp(D - m_g) / s_g) ~ exp(K) (relative 2^-9 + ~2e-6) is then
 * strictly smaller: e^(-1/16) (1 + 2^-9 + 2e-6) < 1 - 2^-9 - 2e-6.
 * This function restates the rule on the CPU (keys, S-th key by sorting, ln in double): tests/test_oracle_golden.py checks on
 * every fixture and on adversarial inputs that the oracle's exact top-S (oracle_group_max_topk over the exact scores) lies
 * inside the candidate set for ANY ctil - zero, exact, perturbed, garbage -, which is all the device needs for identical
 * results.  D [B][G][N] bf16 logits; fin_m / fin_inv [B][G] the finals; ctil [B][G]; cand [B][N] out (1 = candidate);
 * counts [B] out. */
static uint16_t kappa_key(float x) {            /* fixed point, 1/256 per code over [-128, 0): csrc/skv_select_front.h */
    if (!(x > -128.0f)) return 0;
    float t = (x + 128.0f) * 256.0f;
    return (uint16_t)(t >= 32767.0f ? 32767 : (int)t);
}
static float kappa_key_low(int key) {           /* a lower bound of every value carrying `key` (one code of slack for the rounding) */
    return key <= 0 ? -INFINITY : (float)(key - 1) * (1.0f / 256.0f) - 128.0f;
}
static int cmp_u16_desc(const void *a, const void *b) { return (int)*(const uint16_t *)b - (int)*(const uint16_t *)a; }

/* level (per block, nullable): a WITNESS level carried over from the previous step - any key that at least `topk` keys of the
 * row reach serves in place of k15 (the argument only uses "S slots have kappa >= low(level)"); the device tries the carried
 * level first and falls back to the exact S-th key when fewer than S keys reach it (or the candidates exceed its limit). */
ORACLE_API void oracle_fused_candidates(const uint16_t *D, const float *fin_m, const float *fin_inv, const float *ctil,
                                        int blocks, int groups, int n, int topk, const int32_t *level, uint8_t *cand,
                                        int32_t *counts) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < blocks; ++b) {
        uint16_t *key = (uint16_t *)malloc((size_t)n * 2), *srt = (uint16_t *)malloc((size_t)n * 2);
        for (int j = 0; j < n; ++j) {
            float kap = -INFINITY;
            for (int g = 0; g < groups; ++g) {
                float v = bf2f(D[((size_t)b * groups + g) * n + j]) - ctil[b * groups + g];
                if (v > kap) kap = v;
            }
            if (kap > -0x1p-12f) kap = -0x1p-12f;
            key[j] = srt[j] = kappa_key(kap);
        }
        qsort(srt, (size_t)n, 2, cmp_u16_desc);
        int k15 = srt[topk - 1];
        if (level && level[b] >= 8 && level[b] <= k15) k15 = level[b];      /* (level <= S-th key  <=>  at least S keys reach it) */
        double dmax = -1e300, dmin = 1e300;
        for (int g = 0; g < groups; ++g) {
            double c = (double)fin_m[b * groups + g] - log((double)fin_inv[b * groups + g]);
            double d = c - (double)ctil[b * groups + g];
            if (d > dmax) dmax = d;
            if (d < dmin) dmin = d;
        }
        float theta = (float)((double)kappa_key_low(k15) - (dmax - dmin) - 0.0625);
        int thr_lo = theta == theta ? (int)kappa_key(theta) : 0;
        if (thr_lo > k15) thr_lo = k15;
        int c = 0;
        for (int j = 0; j < n; ++j) {
            cand[(size_t)b * n + j] = key[j] >= thr_lo;
            c += key[j] >= thr_lo;
        }
        counts[b] = c;
        free(key);
        free(srt);
    }
}

/* ------------------------------------------------------------------------- */
/* a11: sparse attention over the assembled buffers (models/base.py:341)     */
/* ------------------------------------------------------------------------- */
/* q [bs][q_heads][D] bf16 (q_len = 1), k/v [bs][kv_heads][kv_stride rows][D]
 * bf16, first kv_len rows attended.  out [bs][q_heads][D] bf16 =
 * softmax(q.K^T * scale) V in double precision, rounded once.  flash-attn is
 * un-vendored: the device kernel is compared with this at rtol 1e-3 on f32
 * outputs / 1 bf16 ulp (tests), not bit-exactly. */
ORACLE_API void oracle_sparse_attention(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                                        uint16_t *out, float *out_f32, int bs, int q_heads,
                                        int kv_heads, int head_dim, int kv_len, long kv_stride_rows,
                                        float scale) {
    int G = q_heads / kv_heads;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < bs; ++b)
        for (int qh = 0; qh < q_heads; ++qh) {
            int h = qh / G;
            const uint16_t *qp = q + ((size_t)b * q_heads + qh) * head_dim;
            const uint16_t *kp = k + ((size_t)b * kv_heads + h) * (size_t)kv_stride_rows * head_dim;
            const uint16_t *vp = v + ((size_t)b * kv_heads + h) * (size_t)kv_stride_rows * head_dim;
            double *sc = (double *)malloc((size_t)kv_len * sizeof(double));
            double mx = -1e300;
            for (int j = 0; j < kv_len; ++j) {
                double acc = 0;
                for (int d = 0; d < head_dim; ++d)
                    acc += (double)bf2f(qp[d]) * (double)bf2f(kp[(size_t)j * head_dim + d]);
                sc[j] = acc * (double)scale;
                if (sc[j] > mx) mx = sc[j];
            }
            double den = 0;
            for (int j = 0; j < kv_len; ++j) { sc[j] = exp(sc[j] - mx); den += sc[j]; }
            for (int d = 0; d < head_dim; ++d) {
                double acc = 0;
                for (int j = 0; j < kv_len; ++j) acc += sc[j] * (double)bf2f(vp[(size_t)j * head_dim + d]);
                float r = (float)(acc / den);
                if (out) out[((size_t)b * q_heads + qh) * head_dim + d] = f2bf(r);
                if (out_f32) out_f32[((size_t)b * q_heads + qh) * head_dim + d] = r;
            }
            free(sc);
        }
}

/* a11, second mode: the softmax weights rounded to bf16 at the DEVICE KERNELS' rounding points (round 4).
 * The reference's attention is flash-attn (models/base.py:341, un-vendored), which converts P to the input dtype (bf16)
 * before the P.V matrix product; the device passes that run P.V on the MFMA do the same, each relative to the maximum it
 * knows at that moment, and rescale in f32 afterwards.  Which rows are rounded, and against which maximum, is described
 * by two per-row labels (built by the tests from the kernels' launch geometry and the selection's bookkeeping):
 *   grp[bh][row] >= 0 : rounding group of the row; < 0: the row's weight is never rounded (VALU passes)
 *   ord[bh][row]      : step of the row inside its group (rows of one step are rounded against the same maximum)
 *   m_ref(i) = max{ s_j : grp_j == grp_i, ord_j <= ord_i }            (tile attention: one step per tile -> the tile maximum;
 *                                                                       standalone pass: running maximum of a wave's 32-key steps)
 *   weight_i = bf16( expf( s_i - m_ref(i) ) ) * exp( m_ref(i) - M )   for grp_i >= 0,   exp( s_i - M ) otherwise
 *   out      = sum_i weight_i v_i / sum_i exp( s_i - M )              (the kernels sum the UNROUNDED p into l:
 *                                                                       skv_attn_body.h `ps += p0 + p1`, skv_rebuild.hip `l += pk`)
 * with s_i = f32( q . k_i * scale ) and M the row maximum; sums in double, one final rounding.  What is left between
 * this and the device is f32 summation order, the 2-ulp error of the device's fast exp and the rare weight whose bf16
 * rounding flips because of them (the device's score differs from s_i in its last f32 bits: a weight that lies within
 * ~1e-6 relative of a bf16 rounding boundary lands on the neighbouring bf16 value, one ulp = at most 2^-7 relative, away).
 * flip_f32 (nullable, [bs][q_heads][D]): max over the ROUNDED rows of weight_i |v_i[d]| / normaliser - what ONE such flip
 * can move an output by is at most 2^-7 times this. */
ORACLE_API void oracle_sparse_attention_p16(const uint16_t *q, const uint16_t *k, const uint16_t *v, uint16_t *out,
                                            float *out_f32, int bs, int q_heads, int kv_heads, int head_dim,
                                            int kv_len, long kv_stride_rows, float scale, const int32_t *grp,
                                            const int32_t *ord, float *flip_f32) {
    int G = q_heads / kv_heads;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < bs; ++b)
        for (int qh = 0; qh < q_heads; ++qh) {
            int h = qh / G;
            size_t bh = (size_t)b * kv_heads + h;
            const uint16_t *qp = q + ((size_t)b * q_heads + qh) * head_dim;
            const uint16_t *kp = k + bh * (size_t)kv_stride_rows * head_dim;
            const uint16_t *vp = v + bh * (size_t)kv_stride_rows * head_dim;
            const int32_t *gp = grp + bh * (size_t)kv_len, *op = ord + bh * (size_t)kv_len;
            float *sc = (float *)malloc((size_t)kv_len * sizeof(float));
            double *w = (double *)malloc((size_t)kv_len * sizeof(double));
            int ngrp = 0, nord = 1;
            float mx = -INFINITY;
            for (int j = 0; j < kv_len; ++j) {
                double acc = 0;
                for (int d = 0; d < head_dim; ++d)
                    acc += (double)bf2f(qp[d]) * (double)bf2f(kp[(size_t)j * head_dim + d]);
                sc[j] = (float)(acc * (double)scale);
                if (sc[j] > mx) mx = sc[j];
                if (gp[j] >= ngrp) ngrp = gp[j] + 1;
                if (gp[j] >= 0 && op[j] >= nord) nord = op[j] + 1;
            }
            /* m_ref table: maximum per (group, step), then the running maximum over the steps of a group */
            float *mref = (float *)malloc((size_t)(ngrp > 0 ? ngrp : 1) * nord * sizeof(float));
            for (long i = 0; i < (long)ngrp * nord; ++i) mref[i] = -INFINITY;
            for (int j = 0; j < kv_len; ++j)
                if (gp[j] >= 0) {
                    float *m = &mref[(size_t)gp[j] * nord + (op[j] < 0 ? 0 : op[j])];
                    if (sc[j] > *m) *m = sc[j];
                }
            for (int g = 0; g < ngrp; ++g)
                for (int o = 1; o < nord; ++o)
                    if (mref[(size_t)g * nord + o - 1] > mref[(size_t)g * nord + o]) mref[(size_t)g * nord + o] = mref[(size_t)g * nord + o - 1];
            double den = 0;
            for (int j = 0; j < kv_len; ++j) {
                den += exp((double)sc[j] - (double)mx);
                if (gp[j] >= 0) {
                    float m = mref[(size_t)gp[j] * nord + (op[j] < 0 ? 0 : op[j])];
                    w[j] = (double)bf2f(f2bf(expf(sc[j] - m))) * exp((double)m - (double)mx);
                } else {
                    w[j] = exp((double)sc[j] - (double)mx);
                }
            }
            for (int d = 0; d < head_dim; ++d) {
                double acc = 0, big = 0;
                for (int j = 0; j < kv_len; ++j) {
                    double t = w[j] * (double)bf2f(vp[(size_t)j * head_dim + d]);
                    acc += t;
                    if (gp[j] >= 0 && fabs(t) > big) big = fabs(t);
                }
                float r = (float)(acc / den);
                if (out) out[((size_t)b * q_heads + qh) * head_dim + d] = f2bf(r);
                if (out_f32) out_f32[((size_t)b * q_heads + qh) * head_dim + d] = r;
                if (flip_f32) flip_f32[((size_t)b * q_heads + qh) * head_dim + d] = (float)(big / den);
            }
            free(mref);
            free(w);
            free(sc);
        }
}

/* ------------------------------------------------------------------------- */
/* f1: prefill-side chunk statistics (models/kv_cache.py:854-868)            */
/* ------------------------------------------------------------------------- */
/* k [blocks][>= chunks*8 rows][128] bf16 (block_stride elements between blocks), chunk = 8 consecutive rows.
 *   means[b][c][:]  = key_states_roped_ctx.mean(dim=-2)                       (kv_cache.py:854)
 *   min_cos[b][c]   = cosine_similarity(mean.expand, chunk rows, dim=-1).min(dim=-1).values   (kv_cache.py:859-868)
 * with torch's bf16 tensor semantics, one rounding per ATen op (checked against torch on CPU in
 * tests/test_oracle_golden.py):  mean = bf16(sum_f32(rows) / 8);  ||x|| = bf16(sqrt(sum_f32(x^2))) clamped below at
 * bf16(1e-8);  xn = bf16(x / ||x||);  p = bf16(xn1 * xn2);  cos = bf16(sum_f32(p)).
 * The f32 summation ORDER is this file's contract (torch's differs between CPU and GPU builds and is not
 * specified): 128-wide sums = lane l of 16 sums its 8 contiguous elements sequentially from 0, then a balanced
 * tree over the 16 lane partials in natural order (as score_dot above); the 8-row sum of the mean is
 * ((r0+r4)+(r1+r5)) + ((r2+r6)+(r3+r7)).  Sums of 8 bf16 values are exact in f32 unless their exponents spread over
 * more than 16 bits, so the means agree with torch bit for bit in practice; cos values can differ from torch's by
 * one bf16 ulp on a small fraction of rows. */
static float sum128_lane_tree(const float *x) {
    float part[16];
    for (int l = 0; l < 16; ++l) {
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s = s + x[8 * l + j];
        part[l] = s;
    }
    for (int w = 1; w < 16; w <<= 1)
        for (int l = 0; l < 16; l += 2 * w) part[l] = part[l] + part[l + w];
    return part[0];
}

ORACLE_API void oracle_chunk_stats(const uint16_t *k, long block_stride, int blocks, int chunks,
                                   uint16_t *means, uint16_t *min_cos) {
    const float eps = bf2f(f2bf(1e-8f));
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < blocks; ++b)
        for (int c = 0; c < chunks; ++c) {
            const uint16_t *rows = k + (size_t)b * block_stride + (size_t)c * 8 * 128;
            uint16_t *mo = means + ((size_t)b * chunks + c) * 128;
            float m[128], tmp[128], q1[128];
            for (int d = 0; d < 128; ++d) {
                float a[4];
                for (int r = 0; r < 4; ++r) a[r] = bf2f(rows[r * 128 + d]) + bf2f(rows[(r + 4) * 128 + d]);
                float s = (a[0] + a[1]) + (a[2] + a[3]);
                mo[d] = f2bf(s * 0.125f);
                m[d] = bf2f(mo[d]);
                tmp[d] = m[d] * m[d];
            }
            float n1 = bf2f(f2bf(sqrtf(sum128_lane_tree(tmp))));
            if (n1 < eps) n1 = eps;
            for (int d = 0; d < 128; ++d) q1[d] = bf2f(f2bf(m[d] / n1));
            float best = INFINITY;
            for (int r = 0; r < 8; ++r) {
                float x[128];
                for (int d = 0; d < 128; ++d) {
                    x[d] = bf2f(rows[r * 128 + d]);
                    tmp[d] = x[d] * x[d];
                }
                float n2 = bf2f(f2bf(sqrtf(sum128_lane_tree(tmp))));
                if (n2 < eps) n2 = eps;
                for (int d = 0; d < 128; ++d) tmp[d] = bf2f(f2bf(q1[d] * bf2f(f2bf(x[d] / n2))));
                float cs = bf2f(f2bf(sum128_lane_tree(tmp)));
                if (cs < best) best = cs;
            }
            min_cos[(size_t)b * chunks + c] = f2bf(best + 0.0f);   /* +0: one sign for a zero minimum */
        }
}
