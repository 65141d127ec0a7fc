// Sparse decode attention over the assembled buffers [local | outlier | selected | generated]
// (SURVEY.md section 8 row a11).  Replaces the reference's call into flash-attn
// (/root/reference/models/base.py:341, flash_attn_with_kvcache(q[bs,1,Hq,128], k/v[bs,len,Hkv,128])).
//
// q_len == 1, GQA: the G = Hq/Hkv query heads of one KV head share every K/V row, so a
// workgroup streams each K/V row once and serves all G heads.  The KV range of a (batch, kv head)
// is split over `splits` workgroups (flash-decoding): 8 KV heads alone would leave 248 CUs idle.
//   pass 1  grid (splits, bs*Hkv): a 16-lane group owns one key at a time - the 16 lanes hold the
//           256-B K row (16 B each), finish q.k with a 4-step butterfly, keep an online softmax
//           (m, l) per query head and accumulate p*V for their 8 output dims; the 16 groups of the
//           workgroup are merged through LDS; (m, l, acc[128]) per (split, head) go to a workspace.
//   pass 2  grid (bs*Hq): merges the splits, writes bf16.
// In-place layout (skv_rebuild.hip runs pass 1 over the resident rows as a role of the fetch launch and attends every
// miss tile in the workgroup that builds it):
//   skv_attn_merge_kernel  grid (bs*Hq) x 256: merges the records of both.
// HBM-bound: 2 * kv_len * 256 B per (batch, kv head); K/V rows are read exactly once.
// kv_len may come from device memory (kv_len_dev) so the launch sequence is graph-capturable.
#include "../../include/shadowkv_hip.h"
#include "skv_attn_body.h"

#ifndef SKV_ATTN_PV_MAX_PAIRS
#define SKV_ATTN_PV_MAX_PAIRS 64   // G = 4: (batch, kv head) pairs up to which the all-MFMA split pass is used (0: never)
#endif

template <int G, bool LISTED>
__global__ __launch_bounds__(256) void skv_attn_partial_kernel(
    const bf16_t* __restrict__ q,   // [bs][Hq][128]
    const bf16_t* __restrict__ k,   // [bs][Hkv][rows][128]
    const bf16_t* __restrict__ v,
    float* __restrict__ ws,         // [bs*Hkv][G][splits][AT_REC]  (acc[128], m, l)
    const int* __restrict__ kv_len_dev, int kv_len_host, int kv_rows, long long kv_stride_h /*elements*/, int Hkv,
    int splits, float scale,
    // LISTED (resident set larger than the selection): of the region [sparse_start, + resident_rows) only the chunks in
    // slots[bh][0 .. n_slots) are attended
    const int32_t* __restrict__ slots, int n_slots, int sparse_start, int resident_rows) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    // never past the rows a head owns (the reference's view slice [:sparse_end + gen] clamps the same way)
    const int kv_len = min(kv_len_dev ? *kv_len_dev : kv_len_host, kv_rows);
    // Q.K^T on v_mfma_f32_16x16x32_bf16: G = 8 (unless the launcher took the all-MFMA pass), and G = 4 where the launcher
    // routes large batches here (it leaves room for the score tiles in the LDS request)
    if constexpr ((G == 8 || G == 4) && !LISTED)
        skv_attn_partial_body_mfma<G>(q, k, v, ws, kv_len, kv_stride_h, splits, blockIdx.x, blockIdx.y, scale, s_dyn);
    else
        skv_attn_partial_body<G, LISTED>(q, k, v, ws, kv_len, kv_stride_h, splits, splits, blockIdx.x, blockIdx.y, scale, s_dyn,
                                         LISTED ? slots + (size_t)blockIdx.y * n_slots : nullptr, n_slots, sparse_start,
                                         resident_rows);
}

// Both products on the matrix pipe (skv_attn_partial_body_mfma_pv): latency-bound shapes - one sequence, small batches.
template <int G>
__global__ __launch_bounds__(256, 3) void skv_attn_partial_pv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                     const bf16_t* __restrict__ v, float* __restrict__ ws,
                                                                     const int* __restrict__ kv_len_dev, int kv_len_host,
                                                                     int kv_rows, long long kv_stride_h, int splits, float scale) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int kv_len = min(kv_len_dev ? *kv_len_dev : kv_len_host, kv_rows);
    skv_attn_partial_body_mfma_pv<G>(q, k, v, ws, kv_len, kv_stride_h, splits, blockIdx.x, blockIdx.y, scale, s_dyn);
}

__global__ __launch_bounds__(128) void skv_attn_combine_kernel(const float* __restrict__ ws, bf16_t* __restrict__ out,
                                                               int splits) {
    // block = one query head; its `splits` records are contiguous (splits*528 B): pulled into LDS with
    // 16-B loads that are all in flight together (one memory round trip), then merged from LDS.
    extern __shared__ __attribute__((aligned(16))) float s_rec[];  // [splits][AT_REC]
    const int bq = blockIdx.x, d = threadIdx.x;
    const u32x4* src = reinterpret_cast<const u32x4*>(ws + (size_t)bq * splits * AT_REC);
    const int nvec = splits * (AT_REC / 4);
    u32x4 tmp[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (d + k * 128 < nvec) tmp[k] = src[d + k * 128];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (d + k * 128 < nvec) reinterpret_cast<u32x4*>(s_rec)[d + k * 128] = tmp[k];
    __syncthreads();
    float M = -INFINITY;
    for (int s = 0; s < splits; ++s) M = fmaxf(M, s_rec[s * AT_REC + AT_D]);
    float a = 0.f, L = 0.f;
    for (int s = 0; s < splits; ++s) {
        const float* p = s_rec + s * AT_REC;
        float w = (p[AT_D] == -INFINITY) ? 0.f : __expf(p[AT_D] - M);
        a = __builtin_fmaf(p[d], w, a);
        L = __builtin_fmaf(p[AT_D + 1], w, L);
    }
    out[(size_t)bq * AT_D + d] = f2bf(a / L);
}

// Second half of the overlapped attention (in-place layout): the fetch launch left, per query head, `splits` records of
// the split pass over the resident rows and one record per LIVE miss tile (tiles t >= cnt / 8 of `tiles`; skv_rebuild.hip
// attends every tile it builds).  One workgroup per query head merges them.  All loads are issued at kernel entry (the
// dead tiles' stale records are fetched too and dropped: no dependent round trip behind the hit count).
// MAXREC = 64 (budget 2048: 24 + 32 records; static LDS) or 128 (budget 4096, S = 512: 24 + 64 records - the reference's
// 244K regime, test/e2e.py:50-55; the records then need 67.6 KB of dynamic LDS).
#define MRG_MAX_REC 128
template <int MAXREC>
__global__ __launch_bounds__(256) void skv_attn_merge_kernel(const float* __restrict__ ws, const int32_t* __restrict__ cnts,
                                                             bf16_t* __restrict__ out, int G, int splits, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float s_mrg[];
    float* const s_rec = s_mrg;                              // [MAXREC * AT_REC]
    float* const s_wgt = s_rec + MAXREC * AT_REC;            // [MAXREC]
    float (*s_a)[AT_D] = reinterpret_cast<float (*)[AT_D]>(s_wgt + MAXREC);   // [2][AT_D]
    float* const s_l = s_wgt + MAXREC + 2 * AT_D;            // [2]
    const int bq = blockIdx.x, bh = bq / G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrec = splits + tiles;
    const int cnt = cnts[bh];
    const u32x4* src = reinterpret_cast<const u32x4*>(ws + (size_t)bq * nrec * AT_REC);
    const int nvec = nrec * (AT_REC / 4);
    constexpr int NV = (MAXREC * (AT_REC / 4) + 255) / 256;  // 64 records x 33 vectors / 256 threads = 9
    u32x4 tmp[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (tid + k * 256 < nvec) tmp[k] = src[tid + k * 256];
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (tid + k * 256 < nvec) reinterpret_cast<u32x4*>(s_rec)[tid + k * 256] = tmp[k];
    __syncthreads();
    const int t0 = cnt / 8;                                  // first tile with a miss chunk
    // wave 0: weights exp(m_r - M) of the live records (lane l takes records l, l + 64)
    if (wave == 0) {
        float mr[MAXREC / 64];
        float mloc = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXREC / 64; ++k) {
            const int r = lane + 64 * k;
            const bool live = r < nrec && (r < splits || r - splits >= t0);
            mr[k] = live ? s_rec[r * AT_REC + AT_D] : -INFINITY;
            mloc = fmaxf(mloc, mr[k]);
        }
        const float M = wave_max_dpp(mloc);
#pragma unroll
        for (int k = 0; k < MAXREC / 64; ++k) s_wgt[lane + 64 * k] = (mr[k] == -INFINITY) ? 0.f : __expf(mr[k] - M);
    }
    __syncthreads();
    {
        const int d = tid & (AT_D - 1), half = tid >> 7;     // records r = half, half + 2, ...
        float a = 0.f, L = 0.f;
        // (selects, not a branch: the LDS reads of the next records go out while this one is accumulated - as a branch the
        // loop was one LDS round trip per record)
#pragma unroll 8
        for (int r = half; r < nrec; r += 2) {
            const float wg = s_wgt[r];
            const float ar = s_rec[r * AT_REC + d], lr = s_rec[r * AT_REC + AT_D + 1];
            const float a2 = __builtin_fmaf(ar, wg, a), l2 = __builtin_fmaf(lr, wg, L);
            a = wg != 0.f ? a2 : a;                          // dead records may hold anything (also NaN)
            L = wg != 0.f ? l2 : L;
        }
        s_a[half][d] = a;
        if (d == 0) s_l[half] = L;
    }
    __syncthreads();
    if (tid < AT_D) out[(size_t)bq * AT_D + tid] = f2bf((s_a[0][tid] + s_a[1][tid]) / (s_l[0] + s_l[1]));
}

int skv_launch_attn_merge(const void* ws, const int32_t* cnts, void* out, int bs, int Hq, int Hkv, int S, int splits,
                          hipStream_t st) {
    if (Hkv < 1 || Hq % Hkv || S < 8 || S % 8 || splits < 1 || splits + S / 8 > MRG_MAX_REC) return SKV_ERR_UNSUPPORTED;
    if (splits + S / 8 <= 64) {
        const size_t smem = (size_t)(64 * AT_REC + 64 + 2 * AT_D + 4) * sizeof(float);
        hipLaunchKernelGGL(skv_attn_merge_kernel<64>, dim3(bs * Hq), dim3(256), smem, st, (const float*)ws, cnts, (bf16_t*)out,
                           Hq / Hkv, splits, S / 8);
    } else {
        const size_t smem = (size_t)(128 * AT_REC + 128 + 2 * AT_D + 4) * sizeof(float);
        static size_t attr_bytes[64] = {};
        if (skv_ensure_max_lds((const void*)skv_attn_merge_kernel<128>, smem, attr_bytes) != SKV_OK) return SKV_ERR_LAUNCH;
        hipLaunchKernelGGL(skv_attn_merge_kernel<128>, dim3(bs * Hq), dim3(256), smem, st, (const float*)ws, cnts, (bf16_t*)out,
                           Hq / Hkv, splits, S / 8);
    }
    return SKV_OK;
}

extern "C" size_t skv_attn_workspace_bytes(int bs, int Hq, int splits) { return (size_t)bs * Hq * splits * AT_REC * sizeof(float); }

int skv_launch_sparse_attention(const void* q, const void* k, const void* v, void* out, void* ws,
                                const int* kv_len_dev, int kv_len_host, int kv_rows, long long kv_stride_h, int bs, int Hq,
                                int Hkv, int head_dim, int splits, float scale, const int32_t* slots, int n_slots,
                                int sparse_start, int resident_rows, hipStream_t st) {
    if (head_dim != AT_D || Hkv < 1 || Hq % Hkv != 0 || splits < 1 || splits > 62) return SKV_ERR_UNSUPPORTED;   // (combine: 16 x 128 staging vectors per block)
    if (slots && (n_slots < 0 || sparse_start < 0 || resident_rows < 8 * n_slots || sparse_start + resident_rows > kv_rows))
        return SKV_ERR_ARG;
    if (kv_rows < 1 || (long long)kv_rows * AT_D > kv_stride_h || (!kv_len_dev && (kv_len_host < 1 || kv_len_host > kv_rows)))
        return SKV_ERR_ARG;
    const int G = Hq / Hkv;
    dim3 grid(splits, bs * Hkv), block(256);
    // Which body (tools/attn_mfma_probe.hip, profiles/r03_attn_pv_probe.txt; VALU / Q.K^T on MFMA / both on MFMA, us):
    //   G = 4:   8 pairs 6.96 / 6.89 / 5.58    16 pairs 8.94 / 8.88 / 8.13    32 pairs 12.5 / 13.0 / 12.3
    //           64 pairs 20.4 / 21.8 / 19.8   192 pairs 53.2 / 48.6 / 49.7   (pairs = batch x KV heads)
    //   G = 8:   4 pairs 7.36 / 6.59 / 5.12    32 pairs 18.1 / 16.7 / 12.8
    // -> both products on the matrix pipe (P in bf16, like flash-attn) for G = 8 always and for G = 4 up to
    // SKV_ATTN_PV_MAX_PAIRS pairs; larger G = 4 batches (HBM-bound) take Q.K^T on the MFMA, P.V on the VALU.  A slot list
    // (resident set larger than the selection) always takes the VALU body.
    if (!slots && ((G == 4 && bs * Hkv <= SKV_ATTN_PV_MAX_PAIRS) || G == 8)) {
        if (G == 4)
            hipLaunchKernelGGL(skv_attn_partial_pv_kernel<4>, grid, block, SKV_ATTN_PV_LDS_BYTES(4), st, (const bf16_t*)q,
                               (const bf16_t*)k, (const bf16_t*)v, (float*)ws, kv_len_dev, kv_len_host, kv_rows, kv_stride_h,
                               splits, scale);
        else
            hipLaunchKernelGGL(skv_attn_partial_pv_kernel<8>, grid, block, SKV_ATTN_PV_LDS_BYTES(8), st, (const bf16_t*)q,
                               (const bf16_t*)k, (const bf16_t*)v, (float*)ws, kv_len_dev, kv_len_host, kv_rows, kv_stride_h,
                               splits, scale);
        hipLaunchKernelGGL(skv_attn_combine_kernel, dim3(bs * Hq), dim3(128), (size_t)splits * AT_REC * sizeof(float), st,
                           (const float*)ws, (bf16_t*)out, splits);
        return SKV_OK;
    }
    const size_t smem = (size_t)AT_GROUPS * G * (AT_D + 2) * sizeof(float) + (slots ? (size_t)n_slots * sizeof(int) : 0) +
                        ((G == 8 || G == 4) && !slots ? SKV_ATTN_MFMA_LDS_FLOATS * sizeof(float) : 0);
#define SKV_AT_L(GG, LL)                                                                                        \
    do {                                                                                                        \
        static size_t attr_bytes[64] = {};                                                                      \
        if (skv_ensure_max_lds((const void*)skv_attn_partial_kernel<GG, LL>, smem, attr_bytes) != SKV_OK)       \
            return SKV_ERR_LAUNCH;                                                                              \
        hipLaunchKernelGGL((skv_attn_partial_kernel<GG, LL>), grid, block, smem, st, (const bf16_t*)q,          \
                           (const bf16_t*)k, (const bf16_t*)v, (float*)ws, kv_len_dev, kv_len_host, kv_rows,    \
                           kv_stride_h, Hkv, splits, scale, slots, n_slots, sparse_start, resident_rows);       \
    } while (0)
#define SKV_AT(GG)                                                                                              \
    do {                                                                                                        \
        if (slots) SKV_AT_L(GG, true);                                                                          \
        else SKV_AT_L(GG, false);                                                                               \
    } while (0)
    switch (G) {
        case 1: SKV_AT(1); break;
        case 2: SKV_AT(2); break;
        case 4: SKV_AT(4); break;
        case 8: SKV_AT(8); break;
        default: return SKV_ERR_UNSUPPORTED;
    }
#undef SKV_AT
#undef SKV_AT_L
    hipLaunchKernelGGL(skv_attn_combine_kernel, dim3(bs * Hq), dim3(128), (size_t)splits * AT_REC * sizeof(float), st,
                       (const float*)ws, (bf16_t*)out, splits);
    return SKV_OK;
}
