// Body of the split (flash-decoding) attention pass, shared by the standalone kernel (skv_attn.hip) and by the
// attention role of the fused fetch kernel (skv_rebuild.hip).  See skv_attn.hip for the algorithm.
//   LISTED: of the sparse region [sparse_start, sparse_start + resident_rows) only the chunks in slots[0 .. n_slots) are
//   attended (8 rows each); the pass runs over a VIRTUAL row range - [0, sparse_start), then the listed chunks, then
//   the rows behind the region - so every split gets the same share of live rows whatever the list holds.  The fused
//   fetch launch lists the surviving (hit) chunks: the miss slots are being written by its other roles and are attended
//   there; a resident set larger than the selection lists the selected slots.
#pragma once
#include "skv_common.h"

#define AT_D 128
#define AT_GROUPS 16  // 16-lane groups per 256-thread workgroup
#define AT_REC 132     // floats per (head, split) record: acc[128], m, l, 2 pad (16-B aligned rows)

template <int G, bool LISTED, int AT_KB = 4 /* keys per 16-lane group and iteration: 2 * AT_KB row loads in flight */>
__device__ __forceinline__ void skv_attn_partial_body(
    const bf16_t* __restrict__ q,   // [bs][Hq][128]
    const bf16_t* __restrict__ k,   // [bs][Hkv][rows][128]
    const bf16_t* __restrict__ v,
    float* __restrict__ ws,         // [bs*Hkv][G][rec_splits][AT_REC]  (acc[128], m, l)
    int kv_len, long long kv_stride_h /*elements*/, int splits /* ranges the rows are cut into */,
    int rec_splits /* records per head in ws (>= splits) */, int split, int bh, float scale, float* s_dyn,
    const int32_t* __restrict__ slots, int n_slots, int sparse_start, int resident_rows) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, grp = wave * 4 + (lane >> 4);
    float (*s_part)[G][AT_D + 2] = reinterpret_cast<float (*)[G][AT_D + 2]>(s_dyn);
    int* s_slots = reinterpret_cast<int*>(s_dyn + AT_GROUPS * G * (AT_D + 2));   // [n_slots] when LISTED
    const int listed_rows = LISTED ? 8 * n_slots : 0;
    const int hole = LISTED ? resident_rows - listed_rows : 0;         // rows of the region that are not attended
    if (LISTED) {
        for (int i = tid; i < n_slots; i += 256) s_slots[i] = slots[i];
        __syncthreads();
        kv_len = kv_len <= sparse_start ? kv_len : sparse_start + listed_rows + max(kv_len - sparse_start - resident_rows, 0);
    }
    const int per = (kv_len + splits - 1) / splits;
    const int k0 = split * per, k1 = min(k0 + per, kv_len);
    auto row_of = [&](int key) __attribute__((always_inline)) -> int {
        if (!LISTED) return key;
        const int rel = key - sparse_start;
        if (rel < 0) return key;
        if (rel < listed_rows) return sparse_start + 8 * s_slots[rel >> 3] + (rel & 7);
        return key + hole;
    };


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
    for (int key0 = k0 + grp; key0 < k1; key0 += AT_GROUPS * AT_KB) {
        u32x4 kr[AT_KB], vr[AT_KB];
        bool alive[AT_KB];
#pragma unroll
        for (int i = 0; i < AT_KB; ++i) {
            const int key = key0 + i * AT_GROUPS;
            alive[i] = key < k1;
            // keys past the range re-read its last row (their weight is exp(-inf) = 0)
            const int kc = row_of(alive[i] ? key : k1 - 1);
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
            const bool live = alive[i];
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
            // m = -inf on the first batch: exp(-inf) = 0 (the first key of every batch is alive, so mn is finite)
            const float corr = __expf(m[g] - mn);
            float lsum = l[g] * corr;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] *= corr;
#pragma unroll
            for (int i = 0; i < AT_KB; ++i) {
                const float p = __expf(sc[i][g] - mn);   // dead keys: exp(-inf) = 0
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
        float* dst = ws + (((size_t)bh * G + g) * rec_splits + split) * AT_REC;
        dst[d] = a;
        if (d == 0) {
            dst[AT_D] = M;
            dst[AT_D + 1] = L;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same pass with Q.K^T on the matrix pipe, used for G = 8 (GLM-4: twice the per-key VALU work of G = 4).  Measured
// against the body above on the same data, grid and records (tools/attn_mfma_probe.hip, profiles/r02_attn_mfma_probe.txt):
// G = 8: 7.97 vs 9.00 us (32 splits), 6.67 vs 7.43 us (64 splits); G = 4: 6.95 vs 6.99 us - a tie, the VALU body stays there.
// P.V stays on the VALU: the MFMA A operand of a P.V product needs the keys on the fragment's k index while the scores come
// out with the keys on rows of the C tile; with G <= 8 useful rows of 16 the reshuffle costs more than the 8 FMAs it replaces.
// Needs 4 x 16 x 17 floats of LDS behind the group partials (SKV_ATTN_MFMA_LDS_FLOATS).
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 at_bf16x8;
typedef __attribute__((ext_vector_type(4))) float at_f32x4;
#define SKV_ATTN_MFMA_LDS_FLOATS (4 * 16 * 17)
// MFMA scores: a wave takes 16 keys per step.  A = K rows (lane (r = l & 15, c = l >> 4) loads K[key r][32 ks + 8 c ..+8],
// 4 k-steps), B = Q^T from registers (column g = l & 15, zero for g >= G), C[key (l >> 4) * 4 + i][g = l & 15].
// The scores go through LDS ([key][g]) so that the 16-lane group that owns a V row finds its G weights; the rest (online
// softmax per group, p * V, group merge, record) is the shipped body's.
template <int G>
__device__ __forceinline__ void skv_attn_partial_body_mfma(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, float* __restrict__ ws, int kv_len,
                                                           long long stride_h, int splits, int split, int bh, float scale,
                                                           float* s_dyn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, grp4 = lane >> 4;
    const int per = (kv_len + splits - 1) / splits, k0 = split * per, k1 = min(k0 + per, kv_len);
    float (*s_part)[G][AT_D + 2] = reinterpret_cast<float (*)[G][AT_D + 2]>(s_dyn);
    float* s_sc = s_dyn + AT_GROUPS * G * (AT_D + 2) + wave * 16 * 17;     // per wave [16 keys][16 g] (+1 pad)
    // B fragments: q_g[32 ks + 8 c + j] * scale for g = sub < G
    at_bf16x8 bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 w = {0u, 0u, 0u, 0u};
        if (sub < G) w = *reinterpret_cast<const u32x4*>(q + ((size_t)bh * G + sub) * AT_D + 32 * ks + 8 * grp4);
        bq[ks] = __builtin_bit_cast(at_bf16x8, w);
    }
    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
    }
    const bf16_t* kb = k + (size_t)bh * stride_h;
    const bf16_t* vb = v + (size_t)bh * stride_h + 8 * sub;
    for (int key0 = k0 + wave * 16; key0 < k1; key0 += 64) {       // 4 waves x 16 keys per step
        // A fragments + the V rows of this wave's 16 keys (4 rows per 16-lane group: keys key0 + grp4 + 4 i)
        const int kr = min(key0 + sub, k1 - 1);
        u32x4 ak[4], vr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ak[ks] = *reinterpret_cast<const u32x4*>(kb + (size_t)kr * AT_D + 32 * ks + 8 * grp4);
#pragma unroll
        for (int i = 0; i < 4; ++i) vr[i] = *reinterpret_cast<const u32x4*>(vb + (size_t)min(key0 + grp4 + 4 * i, k1 - 1) * AT_D);
        at_f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf16x8, ak[ks]), bq[ks], c, 0, 0, 0);
        // C[key grp4 * 4 + i][g = sub] -> LDS [key][g]
#pragma unroll
        for (int i = 0; i < 4; ++i) s_sc[(grp4 * 4 + i) * 17 + sub] = c[i] * scale;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): wave-local LDS hand-over
        float sc[4][G];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = grp4 + 4 * i;
#pragma unroll
            for (int g = 0; g < G; ++g) sc[i][g] = (key0 + key < k1) ? s_sc[key * 17 + g] : -INFINITY;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mn = m[g];
#pragma unroll
            for (int i = 0; i < 4; ++i) mn = fmaxf(mn, sc[i][g]);
            const float corr = (mn == -INFINITY) ? 1.f : __expf(m[g] - mn);
            float lsum = l[g] * corr;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] *= corr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = (sc[i][g] == -INFINITY) ? 0.f : __expf(sc[i][g] - mn);
                lsum += p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[g][2 * j] = __builtin_fmaf(p, bf_lo(vr[i][j]), acc[g][2 * j]);
                    acc[g][2 * j + 1] = __builtin_fmaf(p, bf_hi(vr[i][j]), acc[g][2 * j + 1]);
                }
            }
            l[g] = lsum; m[g] = mn;
        }
    }
    const int grp = wave * 4 + grp4;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s_part[grp][g][8 * sub + j] = acc[g][j];
        if (sub == 0) { s_part[grp][g][AT_D] = m[g]; s_part[grp][g][AT_D + 1] = l[g]; }
    }
    __syncthreads();
    for (int o = tid; o < G * AT_D; o += 256) {
        const int g = o / AT_D, d = o % AT_D;
        float M = -INFINITY;
        for (int r = 0; r < AT_GROUPS; ++r) M = fmaxf(M, s_part[r][g][AT_D]);
        float a = 0.f, L = 0.f;
        for (int r = 0; r < AT_GROUPS; ++r) {
            const float mr = s_part[r][g][AT_D], w = (mr == -INFINITY) ? 0.f : __expf(mr - M);
            a = __builtin_fmaf(s_part[r][g][d], w, a);
            L = __builtin_fmaf(s_part[r][g][AT_D + 1], w, L);
        }
        float* dst = ws + (((size_t)bh * G + g) * splits + split) * AT_REC;
        dst[d] = a;
        if (d == 0) { dst[AT_D] = M; dst[AT_D + 1] = L; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: BOTH products on the matrix pipe (tools/attn_mfma_probe.hip, profiles/r03_attn_pv_probe.txt: 19 % faster than
// the VALU pass at G = 4 and 30 % at G = 8 where the pass is latency / issue bound - one sequence or a small batch; no gain
// at 192 (batch, head) pairs, where it is HBM-bound).  A wave takes 32 keys per step:
//   scores   two 16-key tiles C_t[key 16 t + 4 c + i][g = l & 15] (c = l >> 4, i = register) = K Q^T as above;
//   softmax  per column lane g: running max over the lane's 8 scores and the 4 lane groups (two xor shuffles), p = exp(s - m);
//   P.V      out[g][d] = sum_key p[g][key] V[key][d] as v_mfma_f32_16x16x32_bf16 with A = P^T STRAIGHT from the score
//            registers: X = K Q^T has the key on its rows, so X^T V sums over X's row index and needs no lane movement
//            (A element j of lane (g, c) = p of key 16 (j >> 2) + 4 c + (j & 3)); B = V under the SAME key permutation,
//            read k-major from a wave-private LDS image of the 32 V rows with ds_read_b64_tr_b16 (two 4-row x 16-column
//            blocks per fragment); 8 column blocks of 16 dims -> 8 MFMAs + 16 transposed reads per 32 keys; accumulators
//            O[g = 4 c + i][d = 16 nb + (l & 15)], rescaled per step with exp(m_old - m_new) of ROW g (four shuffles).
// P is rounded to bf16 for the MFMA (as flash-attn, the reference's attention, does): |out - out_f32P| <= 2^-9 times the
// attention-weighted mean of |V| (tests add exactly that term).  LDS: [4 waves][G][130] partials + 4 x 32 x 256 B images.
// ---------------------------------------------------------------------------------------------------------------------
#define SKV_ATTN_PV_LDS_BYTES(G) ((size_t)4 * (G) * (AT_D + 2) * sizeof(float) + 4 * 32 * 256)
__device__ __forceinline__ uint32_t skv_lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ u32x2 skv_ds_read_tr16(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
// byte offset of 16-byte chunk ch of row `row` in a [rows][256 B] image (MI355X guide T10, image (b): conflict-free for the
// row-wise 16-B stores and for the transposed reads)
__device__ __forceinline__ uint32_t skv_v_off(int row, int ch) { return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int G>
__device__ __forceinline__ void skv_attn_partial_body_mfma_pv(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                              const bf16_t* __restrict__ v, float* __restrict__ ws, int kv_len,
                                                              long long stride_h, int splits, int split, int bh, float scale,
                                                              float* s_dyn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, c4 = lane >> 4;
    const int per = (kv_len + splits - 1) / splits, k0 = split * per, k1 = min(k0 + per, kv_len);
    float (*s_part)[G][AT_D + 2] = reinterpret_cast<float (*)[G][AT_D + 2]>(s_dyn);           // [4 waves][G][130]
    unsigned char* s_v = reinterpret_cast<unsigned char*>(s_dyn + 4 * G * (AT_D + 2)) + wave * 32 * 256;   // [32 rows][256 B]
    at_bf16x8 bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 w = {0u, 0u, 0u, 0u};
        if (sub < G) w = *reinterpret_cast<const u32x4*>(q + ((size_t)bh * G + sub) * AT_D + 32 * ks + 8 * c4);
        bq[ks] = __builtin_bit_cast(at_bf16x8, w);
    }
    float m = -INFINITY, lsum = 0.f;                   // of head g = sub (lanes sub >= G: padding columns)
    at_f32x4 o[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) o[nb] = (at_f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t* kb = k + (size_t)bh * stride_h;
    const bf16_t* vb = v + (size_t)bh * stride_h;
    const uint32_t sv_base = skv_lds_addr_of(s_v);
    for (int key0 = k0 + wave * 32; key0 < k1; key0 += 128) {      // 4 waves x 32 keys per step (wave-uniform trip count)
        u32x4 ak[2][4], vr[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kr = min(key0 + 16 * t + sub, k1 - 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ak[t][ks] = *reinterpret_cast<const u32x4*>(kb + (size_t)kr * AT_D + 32 * ks + 8 * c4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)                                  // V rows c4 + 4 i, chunk sub (1 KiB per wave-instruction)
            vr[i] = *reinterpret_cast<const u32x4*>(vb + (size_t)min(key0 + c4 + 4 * i, k1 - 1) * AT_D + 8 * sub);
        at_f32x4 sc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sc[t] = (at_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf16x8, ak[t][ks]), bq[ks], sc[t], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(s_v + skv_v_off(c4 + 4 * i, sub)) = vr[i];
        // online softmax of head g = sub over the step's 32 keys
        float s8[8];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = key0 + 16 * t + 4 * c4 + i;
                s8[4 * t + i] = key < k1 ? sc[t][i] * scale : -INFINITY;
                mx = fmaxf(mx, s8[4 * t + i]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);                                // finite: key0 < k1 is alive
        const float corr = __expf(m - mn);                            // first step: exp(-inf) = 0
        m = mn;
        uint32_t pa[4];
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p0 = __expf(s8[2 * j] - mn), p1 = __expf(s8[2 * j + 1] - mn);
            ps += p0 + p1;
            pa[j] = pack_bf2(p0, p1);
        }
        lsum = lsum * corr + ps;
        // rescale the accumulators: row 4 c + i needs the factor of head g = 4 c + i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float cr = __shfl(corr, 4 * c4 + i, 64);
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) o[nb][i] *= cr;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the wave's V image is written (same-wave LDS order)
        const at_bf16x8 afrag = __builtin_bit_cast(at_bf16x8, (u32x4){pa[0], pa[1], pa[2], pa[3]});
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            // lane 4 q + p of the 16-lane group supplies row r0 + q, columns 4 p .. 4 p + 3 of the block (chunk 2 nb + (p >> 1));
            // every lane issues the read (the transposing read needs EXEC all ones: the loop's trip count is wave-uniform)
            const int qrow = sub >> 2, pp = sub & 3;
            const u32x2 b0 = skv_ds_read_tr16(sv_base + skv_v_off(4 * c4 + qrow, 2 * nb + (pp >> 1)) + 8 * (pp & 1));
            const u32x2 b1 = skv_ds_read_tr16(sv_base + skv_v_off(16 + 4 * c4 + qrow, 2 * nb + (pp >> 1)) + 8 * (pp & 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const at_bf16x8 bfrag = __builtin_bit_cast(at_bf16x8, (u32x4){b0[0], b0[1], b1[0], b1[1]});
            o[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, o[nb], 0, 0, 0);
        }
    }
    // l of head g: the four lane groups hold partial sums
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    // wave partial -> LDS: o[nb][i] = out[g = 4 c + i][d = 16 nb + sub]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = 4 * c4 + i;
        if (g < G) {
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) s_part[wave][g][16 * nb + sub] = o[nb][i];
        }
    }
    if (c4 == 0 && sub < G) { s_part[wave][sub][AT_D] = m; s_part[wave][sub][AT_D + 1] = lsum; }
    __syncthreads();
    for (int oo = tid; oo < G * AT_D; oo += 256) {
        const int g = oo / AT_D, d = oo % AT_D;
        float M = -INFINITY;
        for (int r = 0; r < 4; ++r) M = fmaxf(M, s_part[r][g][AT_D]);
        float a = 0.f, L = 0.f;
        for (int r = 0; r < 4; ++r) {                     // (a wave that had no key in range holds m = -inf, acc = l = 0)
            const float mr = s_part[r][g][AT_D], w = (mr == -INFINITY) ? 0.f : __expf(mr - M);
            a = __builtin_fmaf(s_part[r][g][d], w, a);
            L = __builtin_fmaf(s_part[r][g][AT_D + 1], w, L);
        }
        float* dst = ws + (((size_t)bh * G + g) * splits + split) * AT_REC;
        dst[d] = a;
        if (d == 0) { dst[AT_D] = M; dst[AT_D + 1] = L; }
    }
}
