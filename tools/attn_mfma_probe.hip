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

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

template <int G>
static void run(int Hkv, int kv_len, int splits, int layers) {
    const int rows = 2592, Hq = Hkv * G;
    const size_t per_layer = (size_t)Hkv * rows * AT_D;
    std::vector<uint16_t> hk(per_layer), hq((size_t)Hq * AT_D);
    srand(1);
    for (auto& x : hk) x = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& x : hq) x = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    bf16_t *dk, *dv, *dq; float *ws0, *ws1;
    hipMalloc(&dk, per_layer * layers * 2); hipMalloc(&dv, per_layer * layers * 2); hipMalloc(&dq, hq.size() * 2);
    for (int l = 0; l < layers; ++l) {   // distinct buffers per layer (340 MB for 32 layers > Infinity Cache): HBM-cold like the real step
        hipMemcpy(dk + per_layer * l, hk.data(), per_layer * 2, hipMemcpyHostToDevice);
        hipMemcpy(dv + per_layer * l, hk.data(), per_layer * 2, hipMemcpyHostToDevice);
    }
    hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    const size_t wsb = (size_t)Hq * splits * AT_REC * 4;
    hipMalloc(&ws0, wsb); hipMalloc(&ws1, wsb);
    const size_t smem_a = (size_t)AT_GROUPS * G * (AT_D + 2) * 4, smem_b = smem_a + 4 * 16 * 17 * 4;
    hipFuncSetAttribute((const void*)valu_pass<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_a);
    hipFuncSetAttribute((const void*)mfma_pass<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b);
    const float scale = 1.f / sqrtf(128.f);
    dim3 grid(splits, Hkv), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2] = {0, 0};
    for (int round = 0; round < 5; ++round)            // interleaved rounds in one process
        for (int var = 0; var < 2; ++var) {
            hipEventRecord(e0);
            for (int l = 0; l < layers; ++l) {
                if (var == 0) hipLaunchKernelGGL(valu_pass<G>, grid, block, smem_a, 0, dq, dk + per_layer * l, dv + per_layer * l, ws0, kv_len, (long long)rows * AT_D, splits, scale);
                else hipLaunchKernelGGL(mfma_pass<G>, grid, block, smem_b, 0, dq, dk + per_layer * l, dv + per_layer * l, ws1, kv_len, (long long)rows * AT_D, splits, scale);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (round > 0) ms[var] += t;
        }
    std::vector<float> r0(wsb / 4), r1(wsb / 4);
    hipMemcpy(r0.data(), ws0, wsb, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), ws1, wsb, hipMemcpyDeviceToHost);
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
    printf("G=%d kv_heads=%d kv_len=%d splits=%d (%d workgroups): VALU pass %.2f us, MFMA-QK^T pass %.2f us per launch (4 rounds x %d layers, back-to-back, launch gaps included); max |out_valu - out_mfma| = %.2e (bf16 q.k via MFMA has no per-product scale rounding: both exact f32 products)\n",
           G, Hkv, kv_len, splits, splits * Hkv, ms[0] * 1e3 / (4 * layers), ms[1] * 1e3 / (4 * layers), layers, worst);
    hipFree(dk); hipFree(dv); hipFree(dq); hipFree(ws0); hipFree(ws1);
}

int main() {
    run<4>(8, 2497, 32, 32);     // Llama-3.1-8B: 8 KV heads, G = 4
    run<4>(8, 2497, 24, 32);
    run<8>(4, 2497, 32, 40);     // GLM-4-9B: 4 KV heads, G = 8
    run<8>(4, 2497, 64, 40);
    return 0;
}
