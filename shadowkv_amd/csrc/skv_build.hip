// Prefill-side state builder, first pass over the post-RoPE keys: chunk means
// (landmark candidates) and the per-chunk minimum cosine similarity that picks the outlier chunks
// (ShadowKVCache_CPU.prefill_kv_cache, /root/reference/models/kv_cache.py:854-868).  The reference runs this as a
// chain of ATen ops (mean, two vector norms, two divisions, product, sum, min) that each stream the [L, 128] keys or
// a same-sized temporary through HBM; here K is read ONCE (2 KB per chunk, fully coalesced) and 258 B per chunk are
// written.  torch's bf16 semantics need xn = bf16(x / ||x||) per element.  Both operands are bf16 values (8-bit
// significands a, b in [128, 255]), and a quotient a/b of two such integers is never within 2^-18 (relative) of a bf16
// rounding midpoint (a 9-bit odd significand: |512 a - (2m+1) b 2^s| >= 1 and equality would need b >= 256), so
// x * rcp(||x||) (v_rcp_f32, 1 ulp) rounds to the same bf16 as the IEEE division: 1 multiply instead of a ~12
// instruction division sequence per element, bit parity with the oracle kept (tests/test_gpu_build.py).
//
// One wave per chunk: lane (sub = lane & 15, rg = lane >> 4) holds dims 8 sub .. 8 sub + 7 of rows rg and rg + 4.
// Rounding points and summation order: oracle/shadowkv_oracle.c, oracle_chunk_stats.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"

__global__ __launch_bounds__(256) void skv_chunk_stats_kernel(const bf16_t* __restrict__ k, long long block_stride,
                                                              int chunks, bf16_t* __restrict__ means,
                                                              bf16_t* __restrict__ min_cos) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, rg = lane >> 4;
    const bf16_t* kb = k + (size_t)b * block_stride + 8 * sub;
    const float eps = bfr(1e-8f);
    for (int c = blockIdx.x * 4 + wave; c < chunks; c += gridDim.x * 4) {
        const bf16_t* rows = kb + (size_t)c * 1024;
        const u32x4 ra = *reinterpret_cast<const u32x4*>(rows + rg * 128);
        const u32x4 rb = *reinterpret_cast<const u32x4*>(rows + (rg + 4) * 128);
        float xa[8], xb[8], m[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xa[2 * j] = bf_lo(ra[j]); xa[2 * j + 1] = bf_hi(ra[j]);
            xb[2 * j] = bf_lo(rb[j]); xb[2 * j + 1] = bf_hi(rb[j]);
        }
        float n1s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = xa[j] + xb[j];                 // (r + (r+4))
            a = a + __shfl_xor(a, 16);               // (a0 + a1), (a2 + a3)
            a = a + __shfl_xor(a, 32);
            m[j] = bfr(a * 0.125f);
            n1s = n1s + m[j] * m[j];
        }
        float n1 = bfr(sqrtf(row16_tree_sum(n1s)));
        n1 = n1 < eps ? eps : n1;
        float q1[8], sa = 0.f, sb = 0.f, n2a = 0.f, n2b = 0.f;
        const float r1 = __builtin_amdgcn_rcpf(n1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            q1[j] = bfr(m[j] * r1);
            n2a = n2a + xa[j] * xa[j];
            n2b = n2b + xb[j] * xb[j];
        }
        n2a = bfr(sqrtf(row16_tree_sum(n2a)));
        n2b = bfr(sqrtf(row16_tree_sum(n2b)));
        n2a = n2a < eps ? eps : n2a;
        n2b = n2b < eps ? eps : n2b;
        const float ra2 = __builtin_amdgcn_rcpf(n2a), rb2 = __builtin_amdgcn_rcpf(n2b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sa = sa + bfr(q1[j] * bfr(xa[j] * ra2));
            sb = sb + bfr(q1[j] * bfr(xb[j] * rb2));
        }
        const float ca = bfr(row16_tree_sum(sa)), cb = bfr(row16_tree_sum(sb));
        float best = cb < ca ? cb : ca;
        float o = __shfl_xor(best, 16);
        best = o < best ? o : best;
        o = __shfl_xor(best, 32);
        best = o < best ? o : best;
        if (rg == 0) {
            u32x4 mo;
#pragma unroll
            for (int j = 0; j < 4; ++j) mo[j] = pack_bf2(m[2 * j], m[2 * j + 1]);
            *reinterpret_cast<u32x4*>(means + ((size_t)b * chunks + c) * 128 + 8 * sub) = mo;
        }
        if (lane == 0) min_cos[(size_t)b * chunks + c] = f2bf(best + 0.0f);   // +0: one sign for a zero minimum
    }
}

extern "C" int skv_chunk_stats(const void* k, long long block_stride, int blocks, int chunks, int chunk_size,
                               int head_dim, void* means, void* min_cos, skv_stream_t stream) {
    if (!k || !means || !min_cos || blocks < 1 || chunks < 0) return SKV_ERR_ARG;
    if (chunk_size != 8 || head_dim != 128) return SKV_ERR_UNSUPPORTED;
    if (chunks == 0) return SKV_OK;
    int gx = (chunks + 3) / 4;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(skv_chunk_stats_kernel, dim3(gx, blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)k,
                       block_stride, chunks, (bf16_t*)means, (bf16_t*)min_cos);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}
