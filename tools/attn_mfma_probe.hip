// North-star experiment: the split attention pass with Q.K^T on v_mfma_f32_16x16x32_bf16 (the G = 4 or 8
// query heads of a KV head padded to the 16 columns of the tile) against the shipped VALU pass (skv_attn_body.h), same
// data, same grid, same record format.  P.V stays on the VALU in both: the MFMA A operand of a P.V product needs the keys
// on the fragment's k index while the scores come out with the keys on rows of the C tile; with M = G <= 8 useful rows of
// 16 the reshuffle (LDS or tr-reads of V) costs more instructions than the 8 FMAs per (key, head) it replaces.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I shadowkv_amd/csrc tools/attn_mfma_probe.hip -o tools/attn_mfma_probe.bin
#include "../shadowkv_amd/csrc/skv_attn_body.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int G>
__global__ __launch_bounds__(256) void valu_pass(const bf16_t* q, const bf16_t* k, const bf16_t* v, float* ws, int kv_len,
                                                 long long stride_h, int splits, float scale) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    skv_attn_partial_body<G, false>(q, k, v, ws, kv_len, stride_h, splits, splits, blockIdx.x, blockIdx.y, scale, s_dyn, nullptr,
                                    0, 0, 0);
}

// MFMA scores: a wave takes 16 keys per step.  A = K rows (lane (r = l & 15, c = l >> 4) loads K[key r][32 ks + 8 c ..+8],
// 4 k-steps), B = Q^T from registers (column g = l & 15, zero for g >= G), C[key (l >> 4) * 4 + i][g = l & 15].
// The scores go through LDS ([key][g]) so that the 16-lane group that owns a V row finds its G weights; the rest (online
// softmax per group, p * V, group merge, record) is the shipped body's.
template <int G>
__global__ __launch_bounds__(256) void mfma_pass(const bf16_t* q, const bf16_t* k, const bf16_t* v, float* ws, int kv_len,
                                                 long long stride_h, int splits, float scale) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, grp4 = lane >> 4;
    const int bh = blockIdx.y, split = blockIdx.x;
    const int per = (kv_len + splits - 1) / splits, k0 = split * per, k1 = min(k0 + per, kv_len);
    float (*s_part)[G][AT_D + 2] = reinterpret_cast<float (*)[G][AT_D + 2]>(s_dyn);
    float* s_sc = s_dyn + AT_GROUPS * G * (AT_D + 2) + wave * 16 * 17;     // per wave [16 keys][16 g] (+1 pad)
    // B fragments: q_g[32 ks + 8 c + j] * scale for g = sub < G
    bf16x8 bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 w = {0u, 0u, 0u, 0u};
        if (sub < G) w = *reinterpret_cast<const u32x4*>(q + ((size_t)bh * G + sub) * AT_D + 32 * ks + 8 * grp4);
        bq[ks] = __builtin_bit_cast(bf16x8, w);
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
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ak[ks]), bq[ks], c, 0, 0, 0);
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
// Round 3: P.V on the matrix pipe as well.  A wave takes 32 keys per step:
//   scores   two 16-key tiles C_t[key 16 t + 4 c + i][g = l & 15] (c = l >> 4, i = register), as in mfma_pass;
//   softmax  per column lane g: running max over the lane's 8 scores and the 4 lane groups (two xor shuffles), p = exp(s - m);
//   P.V      out[g][d] = sum_key p[g][key] V[key][d] as v_mfma_f32_16x16x32_bf16 with A = P^T taken STRAIGHT from the score
//            registers (MI355X guide, "An accumulator tile as the next MFMA's operand": X = K Q^T has the key on its rows, so
//            X^T V sums over X's row index and needs no lane movement): A element j of lane (g, c) = p of key
//            16 (j >> 2) + 4 c + (j & 3); B = V with the SAME key permutation, read from a wave-private LDS image of the 32 V
//            rows with ds_read_b64_tr_b16 (two 4-row x 16-column blocks per fragment: rows 4c.. of either tile); 8 column
//            blocks of 16 dims -> 8 MFMAs and 16 transposed reads per 32 keys, accumulators O[g = 4 c + i][d = 16 nb + (l & 15)].
//   The accumulator rescale needs exp(m_old - m_new) of ROW g on the lanes that hold row g: four shuffles per step.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ u32x2 ds_read_tr16(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
// byte offset of 16-byte chunk ch of row `row` in a [rows][256 B] image (guide T10, image (b): conflict-free for the
// row-wise 16-B stores and for the transposed reads)
__device__ __forceinline__ uint32_t v_off(int row, int ch) { return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int G>
__global__ __launch_bounds__(256, 3) void mfma_pv_pass(const bf16_t* q, const bf16_t* k, const bf16_t* v, float* ws, int kv_len,
                                                    long long stride_h, int splits, float scale) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, c4 = lane >> 4;
    const int bh = blockIdx.y, split = blockIdx.x;
    const int per = (kv_len + splits - 1) / splits, k0 = split * per, k1 = min(k0 + per, kv_len);
    float (*s_part)[G][AT_D + 2] = reinterpret_cast<float (*)[G][AT_D + 2]>(s_dyn);           // [4 waves][G][130]
    unsigned char* s_v = reinterpret_cast<unsigned char*>(s_dyn + 4 * G * (AT_D + 2)) + wave * 32 * 256;   // [32 rows][256 B]
    bf16x8 bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 w = {0u, 0u, 0u, 0u};
        if (sub < G) w = *reinterpret_cast<const u32x4*>(q + ((size_t)bh * G + sub) * AT_D + 32 * ks + 8 * c4);
        bq[ks] = __builtin_bit_cast(bf16x8, w);
    }
    float m = -INFINITY, lsum = 0.f;                   // of head g = sub (lanes sub >= G: padding columns)
    f32x4 o[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) o[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t* kb = k + (size_t)bh * stride_h;
    const bf16_t* vb = v + (size_t)bh * stride_h;
    const uint32_t sv_base = lds_addr_of(s_v);
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
        f32x4 sc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ak[t][ks]), bq[ks], sc[t], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(s_v + v_off(c4 + 4 * i, sub)) = vr[i];
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
        const bf16x8 afrag = __builtin_bit_cast(bf16x8, (u32x4){pa[0], pa[1], pa[2], pa[3]});
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            // lane 4 q + p of the 16-lane group supplies row r0 + q, columns 4 p .. 4 p + 3 of the block (chunk 2 nb + (p >> 1))
            const int qrow = sub >> 2, pp = sub & 3;
            const u32x2 b0 = ds_read_tr16(sv_base + v_off(4 * c4 + qrow, 2 * nb + (pp >> 1)) + 8 * (pp & 1));
            const u32x2 b1 = ds_read_tr16(sv_base + v_off(16 + 4 * c4 + qrow, 2 * nb + (pp >> 1)) + 8 * (pp & 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, (u32x4){b0[0], b0[1], b1[0], b1[1]});
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
        for (int r = 0; r < 4; ++r) {
            const float mr = s_part[r][g][AT_D], w = (mr == -INFINITY) ? 0.f : __expf(mr - M);
            a = __builtin_fmaf(mr == -INFINITY ? 0.f : s_part[r][g][d], w, a);
            L = __builtin_fmaf(mr == -INFINITY ? 0.f : s_part[r][g][AT_D + 1], w, L);
        }
        float* dst = ws + (((size_t)bh * G + g) * splits + split) * AT_REC;
        dst[d] = a;
        if (d == 0) { dst[AT_D] = M; dst[AT_D + 1] = L; }
    }
}

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

template <int G>
static void run(int Hkv, int kv_len, int splits, int layers) {     // Hkv counts (batch, kv head) pairs
    const int rows = 2592, Hq = Hkv * G;
    const size_t per_layer = (size_t)Hkv * rows * AT_D;
    std::vector<uint16_t> hk(per_layer), hq((size_t)Hq * AT_D);
    srand(1);
    for (auto& x : hk) x = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& x : hq) x = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    bf16_t *dk, *dv, *dq; float *ws0, *ws1, *ws2;
    hipMalloc(&dk, per_layer * layers * 2); hipMalloc(&dv, per_layer * layers * 2); hipMalloc(&dq, hq.size() * 2);
    for (int l = 0; l < layers; ++l) {   // distinct buffers per layer (340 MB for 32 layers > Infinity Cache): HBM-cold like the real step
        hipMemcpy(dk + per_layer * l, hk.data(), per_layer * 2, hipMemcpyHostToDevice);
        hipMemcpy(dv + per_layer * l, hk.data(), per_layer * 2, hipMemcpyHostToDevice);
    }
    hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    const size_t wsb = (size_t)Hq * splits * AT_REC * 4;
    hipMalloc(&ws0, wsb); hipMalloc(&ws1, wsb); hipMalloc(&ws2, wsb);
    const size_t smem_a = (size_t)AT_GROUPS * G * (AT_D + 2) * 4, smem_b = smem_a + 4 * 16 * 17 * 4;
    const size_t smem_c = (size_t)4 * G * (AT_D + 2) * 4 + 4 * 32 * 256;
    hipFuncSetAttribute((const void*)mfma_pv_pass<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_c);
    hipFuncSetAttribute((const void*)valu_pass<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_a);
    hipFuncSetAttribute((const void*)mfma_pass<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b);
    const float scale = 1.f / sqrtf(128.f);
    dim3 grid(splits, Hkv), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3] = {0, 0, 0};
    for (int round = 0; round < 5; ++round)            // interleaved rounds in one process
        for (int var = 0; var < 3; ++var) {
            hipEventRecord(e0);
            for (int l = 0; l < layers; ++l) {
                if (var == 0) hipLaunchKernelGGL(valu_pass<G>, grid, block, smem_a, 0, dq, dk + per_layer * l, dv + per_layer * l, ws0, kv_len, (long long)rows * AT_D, splits, scale);
                else if (var == 1) hipLaunchKernelGGL(mfma_pass<G>, grid, block, smem_b, 0, dq, dk + per_layer * l, dv + per_layer * l, ws1, kv_len, (long long)rows * AT_D, splits, scale);
                else hipLaunchKernelGGL(mfma_pv_pass<G>, grid, block, smem_c, 0, dq, dk + per_layer * l, dv + per_layer * l, ws2, kv_len, (long long)rows * AT_D, splits, scale);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (round > 0) ms[var] += t;
        }
    std::vector<float> r0(wsb / 4), r1(wsb / 4), r2(wsb / 4);
    hipMemcpy(r0.data(), ws0, wsb, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), ws1, wsb, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), ws2, wsb, hipMemcpyDeviceToHost);
    double worst_pv = 0, mag = 0;
    for (size_t rec = 0; rec < (size_t)Hq * splits; ++rec) {
        const float *a = &r0[rec * AT_REC], *b = &r2[rec * AT_REC];
        for (int d = 0; d < AT_D; ++d) {
            const double u = a[d] / a[AT_D + 1], w = b[d] / b[AT_D + 1];
            worst_pv = fmax(worst_pv, fabs(u - w));
            mag = fmax(mag, fabs(u));
        }
    }
    // compare normalised outputs acc / l per record
    double worst = 0;
    for (size_t rec = 0; rec < (size_t)Hq * splits; ++rec) {
        const float *a = &r0[rec * AT_REC], *b = &r1[rec * AT_REC];
        for (int d = 0; d < AT_D; ++d) {
            double x = a[d] / a[AT_D + 1] * exp(a[AT_D] - fmax(a[AT_D], b[AT_D])), y = b[d] / b[AT_D + 1] * exp(b[AT_D] - fmax(a[AT_D], b[AT_D]));
            (void)x; (void)y;
            double u = a[d] / a[AT_D + 1], w = b[d] / b[AT_D + 1];
            worst = fmax(worst, fabs(u - w));
        }
    }
    printf("G=%d kv_heads=%d kv_len=%d splits=%d (%d workgroups): VALU pass %.2f us, MFMA-QK^T pass %.2f us, MFMA-QK^T + MFMA-PV pass %.2f us per launch "
           "(4 rounds x %d layers, back-to-back, launch gaps included); max |out_valu - out_mfma_qk| = %.2e, max |out_valu - out_mfma_qk_pv| = %.2e (max |out| %.2e; P rounded to bf16 for the matrix pipe)\n",
           G, Hkv, kv_len, splits, splits * Hkv, ms[0] * 1e3 / (4 * layers), ms[1] * 1e3 / (4 * layers), ms[2] * 1e3 / (4 * layers), layers, worst, worst_pv, mag);
    hipFree(dk); hipFree(dv); hipFree(dq); hipFree(ws0); hipFree(ws1); hipFree(ws2);
}

int main() {
    run<4>(8, 2497, 32, 32);     // Llama-3.1-8B: 8 KV heads, G = 4
    run<4>(8, 2497, 24, 32);
    run<8>(4, 2497, 32, 40);     // GLM-4-9B: 4 KV heads, G = 8
    run<8>(4, 2497, 64, 40);
    // the standalone pass of a batch, where attention is NOT hidden behind PCIe: bs 24 x 8 KV heads (Llama), bs 8 x 4 (GLM)
    run<4>(192, 2497, 2, 4);
    run<8>(32, 2497, 8, 8);
    // Llama batches 2 / 4 / 8 (tensor_op.sparse_attention_decode picks 256 / (bs * 8) splits)
    run<4>(16, 2497, 16, 16);
    run<4>(32, 2497, 8, 8);
    run<4>(64, 2497, 4, 8);
    return 0;
}
