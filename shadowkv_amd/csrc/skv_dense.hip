// Small fused host-model ops of the decode step (what sits between
// the dense projections and the ShadowKV kernels in LLM.layer_compute, models/base.py:315-341).  At bs = 1
// each of these is a few KB of data; as separate PyTorch launches they cost ~70 us per layer, more than the
// landmark scan.  Memory/latency-bound: one launch each, 16-B accesses where rows allow.
//
//   skv_qkv_rope_update  : split the fused QKV projection, rotate q and k at the token's position, write q
//                          out and push k / v straight into the cache row (replaces vllm rotary_embedding at
//                          llama.py:296 + ShadowKVCache_CPU.update_kv_cache, kv_cache.py:1227-1271)
//   skv_add_rmsnorm      : residual add (bf16) + RMSNorm (flashinfer.norm.rmsnorm, tensor_op.py:34-39)
//   skv_silu_and_mul     : vllm._custom_ops.silu_and_mul (llama.py:421)
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"

// qkv [bs][(Hq + 2*Hkv) * 128] (q_len == 1).  One 64-thread block per (batch, head); thread t owns the
// NeoX pair (t, t+64) or, GLM, t < 32 the interleaved pair (2t, 2t+1) and t >= 32 the pass-through dims.
// bf16 arithmetic with one rounding per operation, like the RoPE kernels of skv_rope.hip.
template <bool GLM>
__global__ __launch_bounds__(64) void skv_qkv_rope_update_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ cos_sin, const int64_t* __restrict__ pos /*[bs]*/,
    const int64_t* __restrict__ row_idx /*[1]*/, const bf16_t* __restrict__ q_override /*nullable [bs][Hq][128]*/,
    bf16_t* __restrict__ q_out /*[bs][Hq][128]*/, bf16_t* __restrict__ k_cache, bf16_t* __restrict__ v_cache,
    int Hq, int Hkv, long long cs_stride, int cs_rows, long long cache_stride_b, long long cache_stride_h, int cache_rows) {
    const int b = blockIdx.y, head = blockIdx.x, t = threadIdx.x;
    const bf16_t* x = qkv + ((size_t)b * (Hq + 2 * Hkv) + head) * 128;
    const long long row = *row_idx;
    if (head >= Hq + Hkv) {  // V: plain copy into the cache row
        if (row >= 0 && row < cache_rows) {
            bf16_t* dst = v_cache + b * cache_stride_b + (long long)(head - Hq - Hkv) * cache_stride_h + row * 128;
            dst[t] = x[t];
            dst[t + 64] = x[t + 64];
        }
        return;
    }
    // the host refuses to step past the table (DecoderLM / GraphDecoder); the clamp keeps a stale counter in bounds
    const long long p = min(max(pos[b], 0ll), (long long)cs_rows - 1);
    const bf16_t* cs = cos_sin + p * cs_stride;
    float o1, o2;
    int i1, i2;
    if (!GLM) {
        i1 = t; i2 = t + 64;
        const float x1 = bf2f(x[i1]), x2 = bf2f(x[i2]), c = bf2f(cs[t]), s = bf2f(cs[t + 64]);
        o1 = bfr(x1 * c) + bfr(-x2 * s);
        o2 = bfr(x2 * c) + bfr(x1 * s);
    } else if (t < 32) {
        i1 = 2 * t; i2 = 2 * t + 1;
        const float x1 = bf2f(x[i1]), x2 = bf2f(x[i2]), c = bf2f(cs[t]), s = bf2f(cs[t + 32]);
        o1 = bfr(x1 * c) + bfr(-x2 * s);
        o2 = bfr(x2 * c) + bfr(x1 * s);
    } else {
        i1 = 32 + t; i2 = 64 + t;  // dims 64..127 pass through
        o1 = bf2f(x[i1]);
        o2 = bf2f(x[i2]);
    }
    if (head < Hq) {
        bf16_t* dst = q_out + ((size_t)b * Hq + head) * 128;
        if (q_override) {  // bench: synthetic selection query, keeps the data dependency on the projection
            const bf16_t* qo = q_override + ((size_t)b * Hq + head) * 128;
            o1 = bf2f(qo[i1]) + o1 * 0.0f;
            o2 = bf2f(qo[i2]) + o2 * 0.0f;
        }
        dst[i1] = f2bf(o1);
        dst[i2] = f2bf(o2);
    } else if (row >= 0 && row < cache_rows) {
        bf16_t* dst = k_cache + b * cache_stride_b + (long long)(head - Hq) * cache_stride_h + row * 128;
        dst[i1] = f2bf(o1);
        dst[i2] = f2bf(o2);
    }
}

// h = x + residual (bf16, one rounding; residual == nullptr: h = x); y = bf16(f32(h) * rsqrt(mean(h^2) + eps) * w).
// One 256-thread block per row; hidden % 8 == 0 and hidden <= 256 * 8 * 4.
__global__ __launch_bounds__(256) void skv_add_rmsnorm_kernel(const bf16_t* __restrict__ x,
                                                              const bf16_t* __restrict__ residual,
                                                              const bf16_t* __restrict__ w, bf16_t* __restrict__ h_out,
                                                              bf16_t* __restrict__ y, int hidden, float eps) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float s_part[4];
    const int nvec = hidden / 8;
    float hv[4][8];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = tid + it * 256;
        if (i < nvec) {
            u32x4 a = reinterpret_cast<const u32x4*>(x + (size_t)r * hidden)[i];
            u32x4 o = a;
            if (residual) {
                u32x4 c = reinterpret_cast<const u32x4*>(residual + (size_t)r * hidden)[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = pack_bf2(bf_lo(a[j]) + bf_lo(c[j]), bf_hi(a[j]) + bf_hi(c[j]));
            }
            if (h_out) reinterpret_cast<u32x4*>(h_out + (size_t)r * hidden)[i] = o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hv[it][2 * j] = bf_lo(o[j]);
                hv[it][2 * j + 1] = bf_hi(o[j]);
                ss = __builtin_fmaf(hv[it][2 * j], hv[it][2 * j], ss);
                ss = __builtin_fmaf(hv[it][2 * j + 1], hv[it][2 * j + 1], ss);
            }
        }
    }
    ss = wave_tree_sum(ss);
    if (lane == 0) s_part[wave] = ss;
    __syncthreads();
    const float tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const float rstd = 1.0f / sqrtf(tot / (float)hidden + eps);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = tid + it * 256;
        if (i < nvec) {
            u32x4 g = reinterpret_cast<const u32x4*>(w)[i];
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf2(hv[it][2 * j] * rstd * bf_lo(g[j]), hv[it][2 * j + 1] * rstd * bf_hi(g[j]));
            reinterpret_cast<u32x4*>(y + (size_t)r * hidden)[i] = o;
        }
    }
}

// out[r][i] = bf16( bf16(silu(x[r][i])) * x[r][inter + i] ), 8 elements per thread
__global__ __launch_bounds__(256) void skv_silu_and_mul_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                               int inter, long long total_vec) {
    const int nvec = inter / 8;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total_vec; v += (long long)gridDim.x * 256) {
        const long long r = v / nvec;
        const int i = (int)(v % nvec);
        u32x4 g = reinterpret_cast<const u32x4*>(x + r * 2 * inter)[i];
        u32x4 u = reinterpret_cast<const u32x4*>(x + r * 2 * inter + inter)[i];
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g0 = bf_lo(g[j]), g1 = bf_hi(g[j]);
            const float s0 = bfr(g0 / (1.0f + __expf(-g0))), s1 = bfr(g1 / (1.0f + __expf(-g1)));
            o[j] = pack_bf2(s0 * bf_lo(u[j]), s1 * bf_hi(u[j]));
        }
        reinterpret_cast<u32x4*>(out + r * inter)[i] = o;
    }
}

// update_kv_cache for a handful of new tokens (/root/reference/models/kv_cache.py:1227-1271): rows [row0, row0 + incoming)
// of both cache buffers <- the new K / V rows; rows past the buffer are dropped like the reference's zero-length slice.
// One 64-thread block (one wave) per (batch, head, new token): threads 0..15 x 2 buffers move 16 B each.
__global__ __launch_bounds__(64) void skv_append_kv_kernel(const u32x4* __restrict__ k_new, const u32x4* __restrict__ v_new,
                                                          u32x4* __restrict__ k_buf, u32x4* __restrict__ v_buf,
                                                          long long k_stride_b, long long k_stride_h, long long k_stride_s,
                                                          long long v_stride_b, long long v_stride_h, long long v_stride_s,
                                                          long long buf_stride_b, long long buf_stride_h, int row0, int rows) {
    const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int row = row0 + t;
    if (row < 0 || row >= rows || tid >= 32) return;
    const int unit = tid & 15;
    const u32x4* src = tid < 16 ? k_new + (b * k_stride_b + h * k_stride_h + t * k_stride_s) / 8 + unit
                                : v_new + (b * v_stride_b + h * v_stride_h + t * v_stride_s) / 8 + unit;
    u32x4* dst = (tid < 16 ? k_buf : v_buf) + (b * buf_stride_b + h * buf_stride_h) / 8 + (long long)row * 16 + unit;
    *dst = *src;
}

static int finish_launch() { return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH; }

extern "C" {

int skv_qkv_rope_update(const void* qkv, const void* cos_sin, const int64_t* pos, const int64_t* row_idx,
                        const void* q_override, void* q_out, void* k_cache, void* v_cache, int batch_size,
                        int q_heads, int kv_heads, int head_dim, long long cos_sin_stride, int cos_sin_rows,
                        long long cache_stride_b, long long cache_stride_h, int cache_rows, int rope_mode,
                        skv_stream_t stream) {
    if (!qkv || !cos_sin || !pos || !row_idx || !q_out || !k_cache || !v_cache || cos_sin_rows < 1) return SKV_ERR_ARG;
    if (head_dim != 128 || (rope_mode != 1 && rope_mode != 2)) return SKV_ERR_UNSUPPORTED;
    dim3 grid(q_heads + 2 * kv_heads, batch_size);
    if (rope_mode == 1)
        hipLaunchKernelGGL(skv_qkv_rope_update_kernel<false>, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                           (const bf16_t*)cos_sin, pos, row_idx, (const bf16_t*)q_override, (bf16_t*)q_out,
                           (bf16_t*)k_cache, (bf16_t*)v_cache, q_heads, kv_heads, cos_sin_stride, cos_sin_rows,
                           cache_stride_b, cache_stride_h, cache_rows);
    else
        hipLaunchKernelGGL(skv_qkv_rope_update_kernel<true>, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                           (const bf16_t*)cos_sin, pos, row_idx, (const bf16_t*)q_override, (bf16_t*)q_out,
                           (bf16_t*)k_cache, (bf16_t*)v_cache, q_heads, kv_heads, cos_sin_stride, cos_sin_rows,
                           cache_stride_b, cache_stride_h, cache_rows);
    return finish_launch();
}

int skv_add_rmsnorm(const void* x, const void* residual, const void* weight, void* h_out, void* y, int rows,
                    int hidden, float eps, skv_stream_t stream) {
    if (!x || !weight || !y || rows < 1) return SKV_ERR_ARG;
    if (hidden % 8 || hidden > 8192) return SKV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(skv_add_rmsnorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)residual, (const bf16_t*)weight, (bf16_t*)h_out, (bf16_t*)y, hidden, eps);
    return finish_launch();
}

int skv_update_kv_cache(const void* k_new, const void* v_new, void* k_buf, void* v_buf, int batch_size, int heads,
                        int incoming, int head_dim, long long k_stride_b, long long k_stride_h, long long k_stride_s,
                        long long v_stride_b, long long v_stride_h, long long v_stride_s, long long buf_stride_b,
                        long long buf_stride_h, int row0, int buf_rows, skv_stream_t stream) {
    if (!k_new || !v_new || !k_buf || !v_buf || batch_size < 1 || heads < 1 || incoming < 1) return SKV_ERR_ARG;
    if (head_dim != 128) return SKV_ERR_UNSUPPORTED;
    if (((k_stride_b | k_stride_h | k_stride_s | v_stride_b | v_stride_h | v_stride_s | buf_stride_b | buf_stride_h) % 8) ||
        (((size_t)k_new | (size_t)v_new | (size_t)k_buf | (size_t)v_buf) & 15))
        return SKV_ERR_ARG;
    hipLaunchKernelGGL(skv_append_kv_kernel, dim3(incoming, heads, batch_size), dim3(64), 0, (hipStream_t)stream,
                       (const u32x4*)k_new, (const u32x4*)v_new, (u32x4*)k_buf, (u32x4*)v_buf, k_stride_b, k_stride_h,
                       k_stride_s, v_stride_b, v_stride_h, v_stride_s, buf_stride_b, buf_stride_h, row0, buf_rows);
    return finish_launch();
}

int skv_silu_and_mul(const void* x, void* out, int rows, int inter, skv_stream_t stream) {
    if (!x || !out || rows < 1) return SKV_ERR_ARG;
    if (inter % 8) return SKV_ERR_UNSUPPORTED;
    const long long total_vec = (long long)rows * (inter / 8);
    long long g = (total_vec + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(skv_silu_and_mul_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)out, inter, total_vec);
    return finish_launch();
}

}  // extern "C"
