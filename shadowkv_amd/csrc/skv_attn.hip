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
// In-place layout (skv_rebuild.hip runs pass 1 over the resident rows as a role of the fetch launch):
//   skv_attn_finish_kernel  grid (bs*Hq) x 1024: attends the miss rows and merges them with the pass-1 records.
// HBM-bound: 2 * kv_len * 256 B per (batch, kv head); K/V rows are read exactly once.
// kv_len may come from device memory (kv_len_dev) so the launch sequence is graph-capturable.
#include "../../include/shadowkv_hip.h"
#include "skv_attn_body.h"

template <int G>
__global__ __launch_bounds__(256) void skv_attn_partial_kernel(
    const bf16_t* __restrict__ q,   // [bs][Hq][128]
    const bf16_t* __restrict__ k,   // [bs][Hkv][rows][128]
    const bf16_t* __restrict__ v,
    float* __restrict__ ws,         // [bs*Hkv][G][splits][AT_REC]  (acc[128], m, l)
    const int* __restrict__ kv_len_dev, int kv_len_host, int kv_rows, long long kv_stride_h /*elements*/, int Hkv,
    int splits, float scale) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    // never past the rows a head owns (the reference's view slice [:sparse_end + gen] clamps the same way)
    const int kv_len = min(kv_len_dev ? *kv_len_dev : kv_len_host, kv_rows);
    skv_attn_partial_body<G, false>(q, k, v, ws, kv_len, kv_stride_h, splits, splits, blockIdx.x, blockIdx.y, scale, s_dyn,
                                    nullptr, 0, 0, 0);
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

// Second half of the overlapped attention (in-place layout): one workgroup per query head attends the MISS rows
// (the chunks fetched / rebuilt by the launch that also ran the split pass over all other rows), then merges its 16
// group partials with the `rec_splits` records of that pass and writes the head's output.
#define FIN_MAX_REC 30   // records of the split pass merged by the finish kernel (30 * 33 16-B vectors <= 1024 threads)
#define FIN_GROUPS 64   // 16-lane groups per workgroup (1024 threads): ~11 miss rows per group at 33 % misses
__global__ __launch_bounds__(FIN_GROUPS * 16) void skv_attn_finish_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const float* __restrict__ ws, const int32_t* __restrict__ dst_slots, const int32_t* __restrict__ cnts,
    bf16_t* __restrict__ out, int G, int S, long long kv_stride_h, int sparse_start, int rec_splits, float scale) {
    __shared__ int s_slot[1024];
    __shared__ float s_part[FIN_GROUPS][AT_D + 2];
    __shared__ __attribute__((aligned(16))) float s_rec[FIN_MAX_REC * AT_REC];
    // XCD-aware block -> head mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2); the G
    // query heads of one KV head read the same K / V rows, so they get block indices that are equal modulo the number of
    // KV heads (8 at the headline shape -> same XCD, the rows come out of one L2).
    const int nkv = gridDim.x / G;
    const int bq = ((int)blockIdx.x % nkv) * G + (int)blockIdx.x / nkv, bh = bq / G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, grp = wave * 4 + (lane >> 4);
    // the whole destination list of this head is requested before the hit count is known (no dependent round trip)
    for (int i = tid; i < S; i += FIN_GROUPS * 16) s_slot[i] = dst_slots[(size_t)bh * S + i];
    const int cnt = cnts[bh], nm = S - cnt, nkeys = nm * 8;
    // the records of the split pass are requested now (one 16-B load per thread, all in flight together) and parked
    // in LDS after the miss-row loop: the merge never waits on global latency
    const int nvec = rec_splits * (AT_REC / 4);
    u32x4 rec_reg = {0u, 0u, 0u, 0u};
    if (tid < nvec) rec_reg = reinterpret_cast<const u32x4*>(ws + (size_t)bq * rec_splits * AT_REC)[tid];
    float qf[8];
    {
        const u32x4 w = *reinterpret_cast<const u32x4*>(q + (size_t)bq * AT_D + 8 * sub);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qf[2 * j] = bf_lo(w[j]) * scale;
            qf[2 * j + 1] = bf_hi(w[j]) * scale;
        }
    }
    __syncthreads();
    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const bf16_t* kb = k + (size_t)bh * kv_stride_h + 8 * sub;
    const bf16_t* vb = v + (size_t)bh * kv_stride_h + 8 * sub;
    constexpr int KB = 8;   // 16 row loads in flight per lane: the typical miss list (<= 64 * 8 * 2 rows) is two batches
    for (int kk0 = grp; kk0 < nkeys; kk0 += FIN_GROUPS * KB) {
        u32x4 kr[KB], vr[KB];
        bool alive[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int kk = kk0 + i * FIN_GROUPS;
            alive[i] = kk < nkeys;
            const int kc = alive[i] ? kk : kk0;                  // kk0 < nkeys: a valid miss row
            const size_t row = (size_t)sparse_start + (size_t)s_slot[cnt + (kc >> 3)] * 8 + (kc & 7);
            kr[i] = *reinterpret_cast<const u32x4*>(kb + row * AT_D);
            vr[i] = *reinterpret_cast<const u32x4*>(vb + row * AT_D);
        }
        float sc[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_fmaf(qf[2 * j], bf_lo(kr[i][j]), s);
                s = __builtin_fmaf(qf[2 * j + 1], bf_hi(kr[i][j]), s);
            }
            s = row16_tree_sum(s);
            sc[i] = alive[i] ? s : -INFINITY;
        }
        float mn = m;
#pragma unroll
        for (int i = 0; i < KB; ++i) mn = fmaxf(mn, sc[i]);
        const float corr = __expf(m - mn);
        float lsum = l * corr;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= corr;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const float p = __expf(sc[i] - mn);
            lsum += p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] = __builtin_fmaf(p, bf_lo(vr[i][j]), acc[2 * j]);
                acc[2 * j + 1] = __builtin_fmaf(p, bf_hi(vr[i][j]), acc[2 * j + 1]);
            }
        }
        l = lsum;
        m = mn;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s_part[grp][8 * sub + j] = acc[j];
    if (sub == 0) {
        s_part[grp][AT_D] = m;
        s_part[grp][AT_D + 1] = l;
    }
    if (tid < nvec) reinterpret_cast<u32x4*>(s_rec)[tid] = rec_reg;
    __syncthreads();
    // ---- merge the FIN_GROUPS group partials and the rec_splits records (<= 94 contributions), all threads:
    // contribution c: c < FIN_GROUPS -> s_part[c], else record c - FIN_GROUPS
    __shared__ float s_w[FIN_GROUPS + FIN_MAX_REC];       // weight exp(m_c - M) of every contribution
    __shared__ float s_red[4];                            // per-wave maxima, then per-wave sums of w * l
    __shared__ float s_a[8][AT_D];
    const int ncon = FIN_GROUPS + rec_splits;
    {   // threads 0..127 (waves 0 and 1) hold one contribution each; everybody runs the same barriers
        const int c = tid;
        float mc = -INFINITY, lc = 0.f;
        if (c < FIN_GROUPS) { mc = s_part[c][AT_D]; lc = s_part[c][AT_D + 1]; }
        else if (c < ncon) { mc = s_rec[(c - FIN_GROUPS) * AT_REC + AT_D]; lc = s_rec[(c - FIN_GROUPS) * AT_REC + AT_D + 1]; }
        const float mw = wave_max_dpp(mc);
        if (wave < 2 && lane == 0) s_red[wave] = mw;
        __syncthreads();
        const float M = fmaxf(s_red[0], s_red[1]);
        const float w = (mc == -INFINITY) ? 0.f : __expf(mc - M);
        if (c < ncon) s_w[c] = w;
        const float lw = wave_tree_sum(w * lc);
        if (wave < 2 && lane == 0) s_red[2 + wave] = lw;
        __syncthreads();
    }
    {
        const int d = tid & (AT_D - 1), part = tid >> 7;     // 8 parts x 128 dims
        float a = 0.f;
        for (int c = part; c < ncon; c += 8) {
            const float x = c < FIN_GROUPS ? s_part[c][d] : s_rec[(c - FIN_GROUPS) * AT_REC + d];
            a = __builtin_fmaf(x, s_w[c], a);
        }
        s_a[part][d] = a;
    }
    __syncthreads();
    if (tid < AT_D) {
        const float L = s_red[2] + s_red[3];
        float a = ((s_a[0][tid] + s_a[1][tid]) + (s_a[2][tid] + s_a[3][tid])) +
                  ((s_a[4][tid] + s_a[5][tid]) + (s_a[6][tid] + s_a[7][tid]));
        out[(size_t)bq * AT_D + tid] = f2bf(a / L);
    }
}

int skv_launch_attn_finish(const void* q, const void* k, const void* v, const void* ws, const int32_t* dst_slots,
                           const int32_t* cnts, void* out, int bs, int Hq, int Hkv, int S, long long kv_stride_h,
                           int sparse_start, int rec_splits, float scale, hipStream_t st) {
    if (Hkv < 1 || Hq % Hkv || S < 1 || S > 1024 || rec_splits < 1 || rec_splits > FIN_MAX_REC) return SKV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(skv_attn_finish_kernel, dim3(bs * Hq), dim3(FIN_GROUPS * 16), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const float*)ws, dst_slots, cnts, (bf16_t*)out, Hq / Hkv, S, kv_stride_h,
                       sparse_start, rec_splits, scale);
    return SKV_OK;
}

extern "C" size_t skv_attn_workspace_bytes(int bs, int Hq, int splits) { return (size_t)bs * Hq * splits * AT_REC * sizeof(float); }

int skv_launch_sparse_attention(const void* q, const void* k, const void* v, void* out, void* ws,
                                const int* kv_len_dev, int kv_len_host, int kv_rows, long long kv_stride_h, int bs, int Hq,
                                int Hkv, int head_dim, int splits, float scale, hipStream_t st) {
    if (head_dim != AT_D || Hkv < 1 || Hq % Hkv != 0 || splits < 1) return SKV_ERR_UNSUPPORTED;
    if (kv_rows < 1 || (long long)kv_rows * AT_D > kv_stride_h || (!kv_len_dev && (kv_len_host < 1 || kv_len_host > kv_rows)))
        return SKV_ERR_ARG;
    const int G = Hq / Hkv;
    dim3 grid(splits, bs * Hkv), block(256);
    const size_t smem = (size_t)AT_GROUPS * G * (AT_D + 2) * sizeof(float);
#define SKV_AT(GG)                                                                                              \
    do {                                                                                                        \
        static bool attr_set = false;                                                                           \
        if (!attr_set && smem > 64 * 1024) {                                                                    \
            (void)hipFuncSetAttribute((const void*)skv_attn_partial_kernel<GG>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                   \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        hipLaunchKernelGGL((skv_attn_partial_kernel<GG>), grid, block, smem, st, (const bf16_t*)q,              \
                           (const bf16_t*)k, (const bf16_t*)v, (float*)ws, kv_len_dev, kv_len_host, kv_rows,  \
                           kv_stride_h, Hkv, splits, scale);                                                                 \
    } while (0)
    switch (G) {
        case 1: SKV_AT(1); break;
        case 2: SKV_AT(2); break;
        case 4: SKV_AT(4); break;
        case 8: SKV_AT(8); break;
        default: return SKV_ERR_UNSUPPORTED;
    }
#undef SKV_AT
    if (splits > 62) return SKV_ERR_UNSUPPORTED;   // 16 x 128 staging vectors per block
    hipLaunchKernelGGL(skv_attn_combine_kernel, dim3(bs * Hq), dim3(128), (size_t)splits * AT_REC * sizeof(float), st,
                       (const float*)ws, (bf16_t*)out, splits);
    return SKV_OK;
}
