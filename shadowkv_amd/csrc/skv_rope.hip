// Stand-alone RoPE kernels behind the legacy `kernels.shadowkv` entry points
// (/root/reference/kernels/rope_new.cu).  The fused decode path does not use them (RoPE lives in
// the epilogue of skv_rebuild.hip); they exist so the reference's 12-function module surface is
// complete and as the simplest known-answer targets.
//
// Memory-bound elementwise work: every lane moves 16 B of x and 16 B of cos/sin per access,
// bf16 arithmetic with the reference's three roundings per output (rope_new.cu:366-367).
#include "skv_common.h"

// One thread = one 8-element unit of a row.  NeoX: unit c (0..7) pairs [8c,8c+8) with [64+8c, ...).
// GLM : unit c (0..15): c < 8 rotates interleaved pairs, c >= 8 copies.
// MODE 0: one position per token (PID ids); MODE 1: chunk ids + cnts skip + push into the cache;
// MODE 2: chunk ids, output laid out like x (apply_rotary_pos_emb_new_v2).
// sin_full != nullptr: separate full-width cos / sin tables (rope.cu:61-151, cos1/cos2, sin1/sin2).
template <bool GLM, int MODE, typename PID>
__global__ __launch_bounds__(256) void skv_rope_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ cos_sin, const bf16_t* __restrict__ sin_full,
    long long ssin, const PID* __restrict__ position_ids,
    bf16_t* __restrict__ out, const int32_t* __restrict__ cnts, int batch, int heads, int seq_len,
    long long sxb, long long sxh, long long sxs, long long scs, long long spb, long long sph, long long sps,
    long long sob, long long soh, long long sos, int off_start, int off_end, int chunk) {
    const int UNITS = GLM ? 16 : 8;
    const long long total = (long long)batch * heads * seq_len * UNITS;
    for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < total;
         u += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(u % UNITS);
        long long r = u / UNITS;
        const int s = (int)(r % seq_len);
        r /= seq_len;
        const int h = (int)(r % heads), b = (int)(r / heads);
        long long pos;
        bf16_t* orow;
        if (MODE == 1) {
            if (s / chunk < cnts[b * heads + h]) continue;
            if (off_start + s >= off_end) continue;
            pos = (long long)position_ids[b * spb + h * sph + (s / chunk) * sps] * chunk + s % chunk;
            orow = out + b * sob + h * soh + (long long)(off_start + s) * sos;
        } else if (MODE == 2) {
            pos = (long long)position_ids[b * spb + h * sph + (s / chunk) * sps] * chunk + s % chunk;
            orow = out + b * sxb + h * sxh + s * sxs;
        } else {
            pos = (long long)position_ids[b * spb + h * sph + s * sps];
            orow = out + b * sxb + h * sxh + s * sxs;
        }
        const bf16_t* xrow = x + b * sxb + h * sxh + s * sxs;
        const bf16_t* cs = cos_sin + pos * scs;
        if (!GLM) {
            u32x4 x1 = *reinterpret_cast<const u32x4*>(xrow + 8 * c);
            u32x4 x2 = *reinterpret_cast<const u32x4*>(xrow + 64 + 8 * c);
            u32x4 cc = *reinterpret_cast<const u32x4*>(cs + 8 * c);
            u32x4 ss, cc2, ss2;
            if (sin_full) {  // separate tables: first half uses [t], second half uses [t + 64]
                const bf16_t* sn = sin_full + pos * ssin;
                ss = *reinterpret_cast<const u32x4*>(sn + 8 * c);
                cc2 = *reinterpret_cast<const u32x4*>(cs + 64 + 8 * c);
                ss2 = *reinterpret_cast<const u32x4*>(sn + 64 + 8 * c);
            } else {
                ss = *reinterpret_cast<const u32x4*>(cs + 64 + 8 * c);
                cc2 = cc;
                ss2 = ss;
            }
            u32x4 o1, o2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a0 = bf_lo(x1[j]), a1 = bf_hi(x1[j]), b0 = bf_lo(x2[j]), b1 = bf_hi(x2[j]);
                float c0 = bf_lo(cc[j]), c1 = bf_hi(cc[j]), s0 = bf_lo(ss[j]), s1 = bf_hi(ss[j]);
                float d0 = bf_lo(cc2[j]), d1 = bf_hi(cc2[j]), e0 = bf_lo(ss2[j]), e1 = bf_hi(ss2[j]);
                o1[j] = pack_bf2(bfr(a0 * c0) + bfr(-b0 * s0), bfr(a1 * c1) + bfr(-b1 * s1));
                o2[j] = pack_bf2(bfr(b0 * d0) + bfr(a0 * e0), bfr(b1 * d1) + bfr(a1 * e1));
            }
            *reinterpret_cast<u32x4*>(orow + 8 * c) = o1;
            *reinterpret_cast<u32x4*>(orow + 64 + 8 * c) = o2;
        } else {
            u32x4 xv = *reinterpret_cast<const u32x4*>(xrow + 8 * c);
            u32x4 o = xv;
            if (c < 8) {
                u32x2 cc = *reinterpret_cast<const u32x2*>(cs + 4 * c);
                u32x2 ss = *reinterpret_cast<const u32x2*>(cs + 32 + 4 * c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xe = bf_lo(xv[j]), xo = bf_hi(xv[j]);
                    uint32_t cw = cc[j >> 1], sw = ss[j >> 1];
                    float cv = (j & 1) ? bf_hi(cw) : bf_lo(cw);
                    float sv = (j & 1) ? bf_hi(sw) : bf_lo(sw);
                    o[j] = pack_bf2(bfr(xe * cv) + bfr(-xo * sv), bfr(xo * cv) + bfr(xe * sv));
                }
            }
            *reinterpret_cast<u32x4*>(orow + 8 * c) = o;
        }
    }
}

static int rope_grid(long long total_units) {
    long long g = (total_units + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// chunk-id variants: mode 1 = push into the cache below/after cnts, mode 2 = output like x
int skv_launch_rope_chunked(const void* x, const void* cos_sin, const int32_t* pid, void* out, const int32_t* cnts,
                            int batch, int heads, int seq_len, int embed_dim, long long sxb, long long sxh,
                            long long sxs, long long sxe, long long scs, long long spb, long long sph, long long sps,
                            long long sob, long long soh, long long sos, int off_start, int off_end, int half_dim,
                            int chunk, int glm, int mode, hipStream_t st) {
    if (embed_dim != 128 || half_dim != 64 || sxe != 1 || chunk < 1) return SKV_ERR_UNSUPPORTED;
    if ((sxb | sxh | sxs | sob | soh | sos) % 8) return SKV_ERR_ARG;
    if (scs % (glm ? 4 : 8)) return SKV_ERR_ARG;
    if (mode == 1 && cnts == nullptr) return SKV_ERR_ARG;
    const long long total = (long long)batch * heads * seq_len * (glm ? 16 : 8);
    if (total <= 0) return SKV_OK;
#define SKV_ROPE(GLMF, MODEF)                                                                                    \
    hipLaunchKernelGGL((skv_rope_kernel<GLMF, MODEF, int32_t>), dim3(rope_grid(total)), dim3(256), 0, st,         \
                       (const bf16_t*)x, (const bf16_t*)cos_sin, (const bf16_t*)nullptr, 0ll, pid, (bf16_t*)out,  \
                       cnts, batch, heads, seq_len, sxb, sxh, sxs, scs, spb, sph, sps, sob, soh, sos, off_start,  \
                       off_end, chunk)
    if (mode == 1) {
        if (glm) SKV_ROPE(true, 1); else SKV_ROPE(false, 1);
    } else if (mode == 2) {
        if (glm) return SKV_ERR_UNSUPPORTED;
        SKV_ROPE(false, 2);
    } else {
        return SKV_ERR_ARG;
    }
#undef SKV_ROPE
    return SKV_OK;
}

// plain RoPE, int64 position per (b, h, s), output laid out like x.  sin == nullptr: fused
// cos|sin table (apply_rotary_pos_emb_new); else separate full-width tables (apply_rotary_pos_emb).
int skv_launch_rope_plain(const void* x, const void* cos_sin, const void* sin, long long ssin, const int64_t* pid,
                          void* out, int batch, int heads, int seq_len, int embed_dim, long long sxb, long long sxh,
                          long long sxs, long long sxe, long long scs, long long spb, long long sph, long long sps,
                          int half_dim, hipStream_t st) {
    if (embed_dim != 128 || half_dim != 64 || sxe != 1) return SKV_ERR_UNSUPPORTED;
    if (((sxb | sxh | sxs) % 8) || (scs % 8) || (sin && (ssin % 8))) return SKV_ERR_ARG;
    const long long total = (long long)batch * heads * seq_len * 8;
    if (total <= 0) return SKV_OK;
    hipLaunchKernelGGL((skv_rope_kernel<false, 0, int64_t>), dim3(rope_grid(total)), dim3(256), 0, st,
                       (const bf16_t*)x, (const bf16_t*)cos_sin, (const bf16_t*)sin, ssin, pid, (bf16_t*)out,
                       (const int32_t*)nullptr, batch, heads, seq_len, sxb, sxh, sxs, scs, spb, sph, sps, 0ll, 0ll,
                       0ll, 0, 0, 1);
    return SKV_OK;
}
