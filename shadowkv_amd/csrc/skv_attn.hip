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
// HBM-bound: 2 * kv_len * 256 B per (batch, kv head); K/V rows are read exactly once.
// kv_len may come from device memory (kv_len_dev) so the launch sequence is graph-capturable.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"

#define AT_D 128
#define AT_GROUPS 16  // 16-lane groups per 256-thread workgroup
#define AT_REC 132     // floats per (head, split) record: acc[128], m, l, 2 pad (16-B aligned rows)

template <int G>
__global__ __launch_bounds__(256) void skv_attn_partial_kernel(
    const bf16_t* __restrict__ q,   // [bs][Hq][128]
    const bf16_t* __restrict__ k,   // [bs][Hkv][rows][128]
    const bf16_t* __restrict__ v,
    float* __restrict__ ws,         // [bs*Hkv][G][splits][AT_REC]  (acc[128], m, l)
    const int* __restrict__ kv_len_dev, int kv_len_host, long long kv_stride_h /*elements*/, int Hkv, int splits,
    float scale) {
    const int bh = blockIdx.y, split = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, grp = wave * 4 + (lane >> 4);
    const int kv_len = kv_len_dev ? *kv_len_dev : kv_len_host;
    const int per = (kv_len + splits - 1) / splits;
    const int k0 = split * per, k1 = min(k0 + per, kv_len);
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float (*s_part)[G][AT_D + 2] = reinterpret_cast<float (*)[G][AT_D + 2]>(s_dyn);

    float qf[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        u32x4 w = *reinterpret_cast<const u32x4*>(q + ((size_t)bh * G + g) * AT_D + 8 * sub);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qf[g][2 * j] = bf_lo(w[j]) * scale;
            qf[g][2 * j + 1] = bf_hi(w[j]) * scale;
        }
    }
    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
    }
    const bf16_t* kb = k + (size_t)bh * kv_stride_h + 8 * sub;
    const bf16_t* vb = v + (size_t)bh * kv_stride_h + 8 * sub;
    // A 16-lane group takes AT_KB keys per iteration (keys grp + 16*i): 2*AT_KB row loads in flight, the scores of
    // the batch are reduced first, then ONE running-max update / accumulator rescale per batch instead of per key.
    constexpr int AT_KB = 4;
    for (int key0 = k0 + grp; key0 < k1; key0 += AT_GROUPS * AT_KB) {
        u32x4 kr[AT_KB], vr[AT_KB];
#pragma unroll
        for (int i = 0; i < AT_KB; ++i) {
            const int key = key0 + i * AT_GROUPS;
            const int kc = key < k1 ? key : k1 - 1;          // clamp; masked below
            kr[i] = *reinterpret_cast<const u32x4*>(kb + (size_t)kc * AT_D);
            vr[i] = *reinterpret_cast<const u32x4*>(vb + (size_t)kc * AT_D);
        }
        float sc[AT_KB][G];
#pragma unroll
        for (int i = 0; i < AT_KB; ++i) {
            float kf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kf[2 * j] = bf_lo(kr[i][j]);
                kf[2 * j + 1] = bf_hi(kr[i][j]);
            }
            const bool live = key0 + i * AT_GROUPS < k1;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s = __builtin_fmaf(qf[g][j], kf[j], s);
                s = row16_tree_sum(s);
                sc[i][g] = live ? s : -INFINITY;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mn = m[g];
#pragma unroll
            for (int i = 0; i < AT_KB; ++i) mn = fmaxf(mn, sc[i][g]);
            const float corr = __expf(m[g] - mn);              // m = -inf on the first batch: exp(-inf) = 0
            float lsum = l[g] * corr;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] *= corr;
#pragma unroll
            for (int i = 0; i < AT_KB; ++i) {
                const float p = __expf(sc[i][g] - mn);         // masked keys: exp(-inf) = 0
                lsum += p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[g][2 * j] = __builtin_fmaf(p, bf_lo(vr[i][j]), acc[g][2 * j]);
                    acc[g][2 * j + 1] = __builtin_fmaf(p, bf_hi(vr[i][j]), acc[g][2 * j + 1]);
                }
            }
            l[g] = lsum;
            m[g] = mn;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s_part[grp][g][8 * sub + j] = acc[g][j];
        if (sub == 0) {
            s_part[grp][g][AT_D] = m[g];
            s_part[grp][g][AT_D + 1] = l[g];
        }
    }
    __syncthreads();
    // merge the 16 groups: thread handles (g, d) pairs; G*128 outputs + G stats
    for (int o = tid; o < G * AT_D; o += 256) {
        const int g = o / AT_D, d = o % AT_D;
        float M = -INFINITY;
#pragma unroll
        for (int r = 0; r < AT_GROUPS; ++r) M = fmaxf(M, s_part[r][g][AT_D]);
        float a = 0.f, L = 0.f;
#pragma unroll
        for (int r = 0; r < AT_GROUPS; ++r) {
            float mr = s_part[r][g][AT_D];
            float w = (mr == -INFINITY) ? 0.f : __expf(mr - M);
            a = __builtin_fmaf(s_part[r][g][d], w, a);
            L = __builtin_fmaf(s_part[r][g][AT_D + 1], w, L);
        }
        float* dst = ws + (((size_t)bh * G + g) * splits + split) * AT_REC;
        dst[d] = a;
        if (d == 0) {
            dst[AT_D] = M;
            dst[AT_D + 1] = L;
        }
    }
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

extern "C" size_t skv_attn_workspace_bytes(int bs, int Hq, int splits) { return (size_t)bs * Hq * splits * AT_REC * sizeof(float); }

int skv_launch_sparse_attention(const void* q, const void* k, const void* v, void* out, void* ws,
                                const int* kv_len_dev, int kv_len_host, long long kv_stride_h, int bs, int Hq,
                                int Hkv, int head_dim, int splits, float scale, hipStream_t st) {
    if (head_dim != AT_D || Hkv < 1 || Hq % Hkv != 0 || splits < 1) return SKV_ERR_UNSUPPORTED;
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
                           (const bf16_t*)k, (const bf16_t*)v, (float*)ws, kv_len_dev, kv_len_host, kv_stride_h, \
                           Hkv, splits, scale);                                                                 \
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
