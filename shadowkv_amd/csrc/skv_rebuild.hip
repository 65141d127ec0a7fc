// K reconstruction for the chunks that missed the cache:
//     k[b,h,i,:] = RoPE( bf16( U[b, pos(i), 0:R] . SV[b,h,:,0:R]^T ), pos(i) ),   pos(i) = id[i/C]*C + i%C
// written straight into the sparse region of the key cache, for rows of chunks >= cnt[b,h].
//
// Replaces /root/reference/kernels/batch_gather_gemm.cu:193-287 (CUTLASS gather-GEMM, bf16 result
// round-tripped through HBM) + /root/reference/kernels/rope_new.cu:321-411 / :429-534 (RoPE-and-push).
// The reference meant to fuse RoPE into the GEMM epilogue and left it disabled
// (batch_gather_gemm_epilogue.h:589-607); here it IS fused, with the same rounding points: the f32
// accumulator is rounded to bf16 first, then rotated in bf16 arithmetic (three roundings per output).
//
// MI355X mapping: one workgroup = 64 rows (8 chunks) x 128 columns, 4 waves x (16 rows x 128 cols).
//   A (U rows)  : gathered straight from HBM in MFMA fragment shape - lane l reads the 16 B
//                 U[pos(row l&15)][32*ks + 8*(l>>4) ...], all 5 k-steps issued up front;
//                 the 8 rows of a chunk are 2,560 contiguous bytes.
//   B (SV[h])   : 40 KB, staged once per workgroup into LDS with rows padded 320 -> 336 B so the
//                 16-row x 16-B ds_read_b128 fragments are bank-conflict free.
//   v_mfma_f32_16x16x32_bf16, 8 column blocks x 5 k-steps per wave.
//   epilogue    : accumulators -> bf16 -> LDS tile; second pass reads whole rows, applies RoPE with
//                 16-B cos/sin loads and stores 16 B per lane (256-B rows, fully coalesced).
// Whole tiles below cnt*C exit early; rows below cnt*C inside a partial tile are computed but not
// stored (they hold resident chunks).  The legacy entry point (pre-RoPE output buffer) shares the kernel.
#include "skv_common.h"
#include "skv_attn_body.h"

// Phase stamps for tools/rb_stamps.py (diagnostic build only, -DSKV_RB_STAMPS -> libshadowkv_hip_stamps.so; no stamp
// executes in the shipped library).  100 MHz wall clock, one row of 8 per workgroup, a buffer nothing else reads.
#ifdef SKV_RB_STAMPS
__device__ unsigned long long g_rb_stamps[16 * 128 * 8];
#define RB_STAMP(i, leader)                                                                                        \
    do {                                                                                                           \
        if ((int)threadIdx.x == (leader) && by < 16 && bx < 128)                                                   \
            g_rb_stamps[((size_t)by * 128 + bx) * 8 + (i)] = wall_clock64();                                       \
    } while (0)
extern "C" __attribute__((visibility("default"))) int skv_debug_rb_stamps(unsigned long long* out, int clear) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rb_stamps), sizeof(g_rb_stamps)) != hipSuccess) return -1;
    if (clear) {
        static unsigned long long zeros[16 * 128 * 8];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rb_stamps), zeros, sizeof(zeros)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define RB_STAMP(i, leader)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RB_ROWS 64
#define RB_D 128
#define RB_SV_PITCH 336  // bytes per SV row in LDS (320 + 16 pad)
#define RB_OUT_PITCH 272 // bytes per output row in LDS (256 + 16 pad)

__device__ __forceinline__ long long chunk_id_at(const void* ids, int ids64, size_t idx) {
    return ids64 ? ((const int64_t*)ids)[idx] : (long long)((const int32_t*)ids)[idx];
}

// KS = rank / 32 k-steps (5 for rank 160); every loop below has a compile-time trip count so the
// compiler batches the global loads of a phase instead of waiting on each one.
// Optional third role of the launch (in-place layout only): split attention over the rows that are final before this
// launch starts (local, outliers, the chunks selected again - dst_slots[0 .. cnt) lists their slots -, generated tokens),
// on the CUs that the PCIe-bound V fetch leaves idle; the miss rows are attended by the workgroups that build them
// (TileAttn below), skv_attn_merge_kernel merges the records.
struct AttnRole {
    const bf16_t* q;          // [bs][Hq][128]
    float* ws;                // [bs*Hq][rec_splits][AT_REC]
    const int* kv_len_dev;    // nullable
    int kv_len_host, kv_rows, splits, rec_splits;
    int resident_rows;        // rows of the sparse region (resident slots x 8): the generated rows sit behind it
    float scale;
    const short* early_of;    // speculative early V fetch (skv_early.hip; null = off): staging index per chunk id
    const u32x4* early_staging;
    int early_chunks, early_max;
};


// Attention over the 64 rows of ONE freshly built miss tile (in-place layout, attention role present): the workgroup
// that rebuilt the K rows of 8 miss chunks holds them in LDS and holds the chunks' V rows in REGISTERS (the landing loads:
// thread (rsub, unit) has 16 B = 8 dims of token unit / 16 of chunk 2k + rsub, i.e. a 16-lane group holds whole V rows),
// so it attends the tile right there for all G query heads of the KV head and emits one record (acc[128], m, l) per
// head - the miss rows are never read back from HBM by a second pass.
// Round 3: everything that can happen BEFORE the V rows are back happens while the host loads are in flight - q.k for the
// group's 4 rows (VALU, K tile and q in LDS), the tile-wide maximum per head, the softmax weights, written to LDS as the
// bf16 A operand P[g][key] of an MFMA -, so that what remains behind the PCIe round trip is: V registers -> cache and ->
// an LDS image, one barrier, out[g][d] = sum_key P[g][key] V[key][d] as 4 v_mfma_f32_16x16x32_bf16 per landing wave (two
// 16-dim column blocks x two 32-key steps; V comes back k-major through ds_read_b64_tr_b16), and the record.  No per-group
// partials and no merge of 16 of them any more: in-kernel stamps put the part behind the last V row at 1.6 us (G = 4) /
// 3.2 us (G = 8) before (profiles/r03_fetch_stamps.txt).  P is rounded to bf16 for the MFMA as in the standalone pass.
// Rows below `first_live` are resident (hit) rows: weight 0 here (the split pass has them).
// LDS (over the SV staging area, dead once the rebuild waves' MFMA phase is over):
#define TA_V_BYTES (64 * 256)                    // V image [64 rows][256 B], T10 swizzle
#define TA_P_BYTES (16 * 64 * 2)                 // P [16 head rows (>= G: zero)][64 keys] bf16
template <int G>
struct TileAttn {
    float sc[4][G];                              // scores of the group's 4 rows (the 16 lanes of a group hold the same values)

    // phase A1: q . k for the group's 4 rows, group maximum per head -> s_gmax[grp][g].
    // vt = thread index among the 256 attention threads (waves 4..7 of the workgroup)
    __device__ __forceinline__ void scores(const bf16_t* q, const unsigned char* sK, int first_live, float scale, int vt,
                                           float* s_gmax) {
        const int sub = vt & 15, rsub = vt >> 7, g8 = (vt & 127) >> 4, grp = vt >> 4;
        float qf[G][8];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const u32x4 wq = *reinterpret_cast<const u32x4*>(q + (size_t)g * AT_D + 8 * sub);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qf[g][2 * j] = bf_lo(wq[j]) * scale;
                qf[g][2 * j + 1] = bf_hi(wq[j]) * scale;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (2 * k + rsub) * 8 + g8;                 // the row whose V piece this thread loaded as lv[k]
            const bool alive = r >= first_live;
            // dead rows hold the un-rotated product of a resident chunk: never let it meet a weight
            const u32x4 kr = alive ? *reinterpret_cast<const u32x4*>(sK + r * RB_OUT_PITCH + sub * 16) : (u32x4){0u, 0u, 0u, 0u};
            float kf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kf[2 * j] = bf_lo(kr[j]);
                kf[2 * j + 1] = bf_hi(kr[j]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float sd = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) sd = __builtin_fmaf(qf[g][j], kf[j], sd);
                sd = row16_tree_sum(sd);
                sc[k][g] = alive ? sd : -INFINITY;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (sub == g) s_gmax[grp * G + g] = fmaxf(fmaxf(sc[0][g], sc[1][g]), fmaxf(sc[2][g], sc[3][g]));
    }

    // phase A2 (behind a barrier): tile maximum per head, weights relative to it -> P (bf16) and the group's sum -> s_gl
    __device__ __forceinline__ void weights(int vt, const float* s_gmax, bf16_t* s_P, float* s_gl, float* s_M) {
        const int sub = vt & 15, rsub = vt >> 7, g8 = (vt & 127) >> 4, grp = vt >> 4;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float M = -INFINITY;
#pragma unroll
            for (int r = 0; r < AT_GROUPS; ++r) M = fmaxf(M, s_gmax[r * G + g]);      // (finite: the tile has a live row)
            if (sub == g) {                                     // one lane of the group per head
                float l = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float pk = sc[k][g] == -INFINITY ? 0.f : __expf(sc[k][g] - M);
                    l += pk;
                    s_P[g * 64 + (2 * k + rsub) * 8 + g8] = f2bf(pk);
                }
                s_gl[grp * G + g] = l;
                if (grp == 0) s_M[g] = M;
            }
        }
    }
};

// phase B of the tile attention, landing waves only (lw = 0..3), behind the barrier that follows the V image: two 16-dim
// column blocks per wave, out[g][d] over the tile's 64 keys, record (acc[128], m, l) per head.
template <int G>
__device__ __forceinline__ void skv_tile_pv_mfma(const unsigned char* s_v, const bf16_t* s_P, const float* s_gl, const float* s_M,
                                                 float* __restrict__ rec, size_t rec_stride_g, int lane, int lw) {
    typedef __attribute__((ext_vector_type(8))) __bf16 ta_bf16x8;
    typedef __attribute__((ext_vector_type(4))) float ta_f32x4;
    const int sub = lane & 15, c4 = lane >> 4, qrow = sub >> 2, pp = sub & 3;
    const uint32_t sv_base = skv_lds_addr_of(s_v);
    ta_bf16x8 a[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)                               // A[row g = sub][k = key 32 ks + 8 c + j]
        a[ks] = __builtin_bit_cast(ta_bf16x8, *reinterpret_cast<const u32x4*>(s_P + sub * 64 + 32 * ks + 8 * c4));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int nb = 2 * lw + h;
        ta_f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // B[k = key 32 ks + 8 c + j][n = sub]: two 4-row x 16-column blocks, rows 32 ks + 8 c (+ 4) .. ; lane 4 q + p of the
            // 16-lane group supplies row r0 + q, columns 4 p .. 4 p + 3 (chunk 2 nb + (p >> 1))
            const int r0 = 32 * ks + 8 * c4;
            const u32x2 b0 = skv_ds_read_tr16(sv_base + skv_v_off(r0 + qrow, 2 * nb + (pp >> 1)) + 8 * (pp & 1));
            const u32x2 b1 = skv_ds_read_tr16(sv_base + skv_v_off(r0 + 4 + qrow, 2 * nb + (pp >> 1)) + 8 * (pp & 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], __builtin_bit_cast(ta_bf16x8, (u32x4){b0[0], b0[1], b1[0], b1[1]}), o, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                            // o[i] = out[g = 4 c + i][d = 16 nb + sub]
            const int g = 4 * c4 + i;
            if (g < G) rec[(size_t)g * rec_stride_g + 16 * nb + sub] = o[i];
        }
    }
    if (lw == 0 && lane < G) {                                   // m and l of head g = lane
        float L = 0.f;
#pragma unroll
        for (int r = 0; r < AT_GROUPS; ++r) L += s_gl[r * G + lane];
        rec[(size_t)lane * rec_stride_g + AT_D] = s_M[lane];
        rec[(size_t)lane * rec_stride_g + AT_D + 1] = L;
    }
}

template <int MODE /*0 = pre-RoPE output, 1 = Llama RoPE, 2 = GLM RoPE*/, int KS, int AG = 0 /* attention role: G */>
__global__ __launch_bounds__(AG > 0 ? 512 : 256, AG > 0 ? 2 : 1) void skv_rebuild_kernel(
    const bf16_t* __restrict__ U,        // [bs][seq_len][R]
    const bf16_t* __restrict__ SV,       // [bs][heads][128][R]
    const bf16_t* __restrict__ cos_sin,  // [max_pos][cs_stride]
    const void* __restrict__ ids,        // [bs][heads][S] chunk id per slot (int64 or int32)
    const int32_t* __restrict__ cnts,    // [bs*heads] (nullable -> 0)
    bf16_t* __restrict__ out,            // MODE 0: [bs][heads][S*C][128]; else key cache
    int heads, int seq_len, int S, int C, int ids64, long long cs_stride,
    long long out_stride_b, long long out_stride_h, long long out_stride_s, int out_row0,
    const bf16_t* __restrict__ hit_temp, // nullable: [bs*heads][S][C*128] moved hit chunks staged by skv_stage_hits
    const int32_t* __restrict__ hit_offsets /* [bs*heads][S] */,
    // in-place layout (skv_select_chunks_inplace): non-null -> virtual slot j in [cnt, S) is the (j-cnt)-th miss, its
    // chunk id comes from `ids` (the int32 offsets array) and its rows land in slot dst_slots[j]; hits never move
    const int32_t* __restrict__ dst_slots /* [bs*heads][S] nullable */,
    // optional second role (blocks blockIdx.x >= rebuild_tiles): land the V chunks of this (batch, head) -
    // moved hits from v_temp, misses from the pinned host table - so "K rebuild || V fetch" is ONE launch with no
    // stream fork/join around it.  Rebuild tiles come first in dispatch order (short, few), landing blocks fill
    // the remaining CUs and are PCIe-bound.
    int rebuild_tiles, const u32x4* __restrict__ v_host, u32x4* __restrict__ v_buf, const u32x4* __restrict__ v_temp,
    long long v_host_stride_u128, long long v_stride_u128, long long v_off_u128, int land_blocks, AttnRole ar) {
    // block -> (role index bx, batch*head by).  Attention role present: a 1-D grid in which the tile blocks (from the
    // highest tile index down, all heads of a tile index together: the live tiles - miss chunks - are the last indices
    // and are the long pole, their host loads should go out at once) are INTERLEAVED with the split-attention blocks over
    // the resident rows in the ratio of their counts.  One 512-thread workgroup runs per CU (the launcher asks for > 80 KB
    // of LDS; __launch_bounds__(512, 2) = two waves per SIMD = that one workgroup), blocks get CUs in id order:
    // with all tiles first, a batch with more live tiles than CUs (bs >= 3 at 67 % hits) would start the attention blocks
    // only after the PCIe-bound tiles had drained and lose the overlap.
    int bx = blockIdx.x, by = blockIdx.y;
    if constexpr (AG > 0) {
        const int per_head = rebuild_tiles + ar.splits;
        const int nbh = (int)gridDim.x / per_head;
        const long long na = (long long)nbh * ar.splits, nall = (long long)gridDim.x;
        const int a_before = (int)((long long)bx * na / nall);                   // attention blocks with a smaller id
        const bool is_attn = (int)(((long long)bx + 1) * na / nall) > a_before;
        if (!is_attn) {
            const int t = bx - a_before;                                          // tile blocks with a smaller id
            by = t % nbh;
            bx = rebuild_tiles - 1 - t / nbh;
        } else {
            by = a_before % nbh;
            bx = rebuild_tiles + a_before / nbh;
        }
    }
    if constexpr (AG > 0) if (bx >= rebuild_tiles + land_blocks) {
        if (threadIdx.x >= 256) return;                       // the split pass is a 256-thread body
        RB_STAMP(0, 0);
        extern __shared__ __attribute__((aligned(16))) unsigned char smem_a[];
        const int bh3 = by, cnt3 = cnts ? cnts[bh3] : 0;
        const int kv_len = min(ar.kv_len_dev ? *ar.kv_len_dev : ar.kv_len_host, ar.kv_rows);
        skv_attn_partial_body<AG, true, (AG == 8 ? 1 : 2)>(
            ar.q, out, reinterpret_cast<const bf16_t*>(v_buf), ar.ws, kv_len, out_stride_h, ar.splits, ar.rec_splits,
            bx - rebuild_tiles - land_blocks, bh3, ar.scale, reinterpret_cast<float*>(smem_a),
            dst_slots + (size_t)bh3 * S, cnt3, out_row0, ar.resident_rows);
        RB_STAMP(7, 0);
        return;
    }
    if (bx >= rebuild_tiles) {
        const int bh2 = by, tid2 = threadIdx.x, unit = tid2 & 127, rsub = tid2 >> 7;
        const int cnt2 = cnts ? cnts[bh2] : 0;
        const int blk = bx - rebuild_tiles;
        u32x4 lv[4];
        bool lact[4];
        int loffs[4], ldi[4];     // source and destination of the thread's four rows, looked up BEFORE the host loads
#pragma unroll                    // (loads return in order: a lookup issued behind them would wait for the PCIe round trip)
        for (int k = 0; k < 4; ++k) {
            const int i = min(blk * 8 + k * 2 + rsub, S - 1);
            loffs[k] = hit_offsets[(size_t)bh2 * S + i];
            ldi[k] = dst_slots ? dst_slots[(size_t)bh2 * S + i] : i;
        }
        // speculative early V fetch (skv_early.h): chunks already staged in HBM (one more dependent lookup, then no PCIe)
        int lsrc[4] = {-1, -1, -1, -1};
        if (ar.early_of != nullptr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = blk * 8 + k * 2 + rsub;
                const int c = min(max(loffs[k], 0), ar.early_chunks - 1);
                const int e = ar.early_of[(size_t)bh2 * ar.early_chunks + c];
                lsrc[k] = (i < S && i >= cnt2 && loffs[k] == c && e < ar.early_max) ? e : -1;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = blk * 8 + k * 2 + rsub;
            const int off = loffs[k];
            const bool moved_hit = i < S && i < cnt2 && !dst_slots && off != i;
            const bool miss = i < S && i >= cnt2;
            lact[k] = moved_hit || miss;
            // unconditional load through a selected pointer (a load under `if` is followed by vmcnt(0): the four PCIe
            // round trips of a thread would be serialised); inactive rows read a valid, unused row of the cache
            const u32x4* src = miss ? (lsrc[k] >= 0 ? ar.early_staging + (((long long)bh2 * ar.early_max + lsrc[k]) * 128 + unit)
                                                    : v_host + ((long long)bh2 * v_host_stride_u128 + (long long)off * 128 + unit))
                               : moved_hit ? v_temp + (((long long)bh2 * S + i) * 128 + unit)
                                           : v_buf + ((long long)bh2 * v_stride_u128 + v_off_u128 + unit);
            lv[k] = *src;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (lact[k]) v_buf[(long long)bh2 * v_stride_u128 + v_off_u128 + (long long)ldi[k] * 128 + unit] = lv[k];
        }
        return;
    }
    constexpr int R = KS * 32;
    constexpr int UNITS_PER_ROW = R / 8;                       // 16-B units per SV row
    constexpr int SV_ITERS = (RB_D * UNITS_PER_ROW + 255) / 256;
    constexpr int EPI_UNITS = MODE == 1 ? 8 : 16;              // 8-element units per output row
    constexpr int EPI_ITERS = RB_ROWS * EPI_UNITS / 256;       // 2 (Llama) or 4
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sSV = smem;                         // 128 x 336 B
    unsigned char* sOut = smem + RB_D * RB_SV_PITCH;   // 64 x 272 B
    const int bh = by, b = bh / heads, h = bh % heads;
    const int i0 = bx * RB_ROWS;
    const int total_rows = S * C;
    const int cnt = cnts ? cnts[bh] : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (hit_temp != nullptr && C == 8) {
        // second half of the two-phase compaction (skv_move.hip): land the moved hit chunks of this tile
        // (chunk slot j < cnt whose source slot differs) from the staging buffer into their final rows.
        const int j0 = i0 / 8;                       // first chunk slot of the tile (RB_ROWS / C = 8 slots)
        const int unit = tid & 127, rsub = tid >> 7;
        u32x4 hv[4];
        bool hact[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + k * 2 + rsub;
            hact[k] = j < cnt && j < S && hit_offsets[(size_t)bh * S + j] != j;
            if (hact[k]) hv[k] = reinterpret_cast<const u32x4*>(hit_temp)[((size_t)bh * S + j) * 128 + unit];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + k * 2 + rsub;
            if (hact[k])
                reinterpret_cast<u32x4*>(out + (size_t)b * out_stride_b + (size_t)h * out_stride_h +
                                         (size_t)(out_row0 + j * 8) * out_stride_s)[unit] = hv[k];
        }
    }
    if (i0 + RB_ROWS <= cnt * C) return;  // every row of this tile is a resident (hit) row

    // Attention role present (in-place layout, C == 8, S % 8 == 0): waves 4..7 LAND the V chunks of the tile's 8 slots and
    // attend the finished tile (TileAttn) from those registers; waves 0..3 rebuild the K tile below.
    // tile-attention scratch over the SV staging area (dead once the MFMA phase of the rebuild waves is over, barrier (2)):
    // V image | P | group maxima | group sums | tile maxima
    unsigned char* const s_v = smem;
    bf16_t* const s_P = reinterpret_cast<bf16_t*>(smem + TA_V_BYTES);
    float* const s_gmax = reinterpret_cast<float*>(smem + TA_V_BYTES + TA_P_BYTES);
    float* const s_gl = s_gmax + AT_GROUPS * (AG > 0 ? AG : 1);
    float* const s_M = s_gl + AT_GROUPS * (AG > 0 ? AG : 1);
    static_assert(TA_V_BYTES + TA_P_BYTES + (2 * AT_GROUPS + 1) * 8 * sizeof(float) <= (size_t)RB_D * RB_SV_PITCH, "tile-attention scratch");
    const bf16_t* const q_lds = reinterpret_cast<const bf16_t*>(sOut + RB_ROWS * RB_OUT_PITCH);     // [G][128] behind the K tile
    if constexpr (AG > 0) {
        float* const rec = ar.ws + ((size_t)bh * AG * ar.rec_splits + ar.splits + bx) * AT_REC;
        const size_t rec_stride = (size_t)ar.rec_splits * AT_REC;
        if (tid >= 256) {
            const int vt = tid - 256, unit = vt & 127, rsub = vt >> 7, jb = i0 / 8;
            RB_STAMP(0, 256);
            int loff[4], lslot[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = min(jb + 2 * k + rsub, S - 1);
                loff[k] = hit_offsets[(size_t)bh * S + j];        // miss id (source chunk) and destination slot of the
                lslot[k] = dst_slots[(size_t)bh * S + j];         // thread's four chunks; stale for j < cnt, unused then
            }
            // chunks the early launch has staged already (a second, dependent lookup - it returns well inside the wait for
            // barrier (1), which stands behind TWO dependent round trips of the rebuild waves)
            int lsrc[4] = {-1, -1, -1, -1};
            if (ar.early_of != nullptr) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = jb + 2 * k + rsub;
                    const int c = min(max(loff[k], 0), ar.early_chunks - 1);
                    const int e = ar.early_of[(size_t)bh * ar.early_chunks + c];
                    lsrc[k] = (j >= cnt && j < S && loff[k] == c && e < ar.early_max) ? e : -1;
                }
            }
            u32x4 qreg = {0u, 0u, 0u, 0u};
            if (vt < AG * 16) qreg = reinterpret_cast<const u32x4*>(ar.q + (size_t)bh * AG * AT_D)[vt];
            if (vt < AG * 16) *reinterpret_cast<u32x4*>(sOut + RB_ROWS * RB_OUT_PITCH + vt * 16) = qreg;
            // (1) the rebuild waves have RECEIVED every global load they need (U fragments, SV, cos / sin, slots): measured
            // with in-kernel stamps (tools/rb_stamps.py), loads of device memory issued on a CU while host-memory loads of
            // the same CU are outstanding come back only with them (K tile ready at 24 us instead of 7 us) - so the host
            // loads go out behind this barrier, and nothing else on this CU (one 512-thread workgroup per CU) touches
            // global memory until they are back.
            __syncthreads();
            u32x4 lv[4];
            bool lact[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = jb + 2 * k + rsub;
                lact[k] = j >= cnt && j < S;
                // unconditional load through a selected pointer: a load under `if` makes hipcc wait vmcnt(0) right behind
                // it, i.e. one PCIe round trip per chunk; dead chunks read a (valid, unused) row of the device cache instead
                const u32x4* src = !lact[k] ? v_buf + ((long long)bh * v_stride_u128 + v_off_u128 + unit)
                                   : lsrc[k] >= 0 ? ar.early_staging + (((long long)bh * ar.early_max + lsrc[k]) * 128 + unit)
                                                  : v_host + ((long long)bh * v_host_stride_u128 + (long long)loff[k] * 128 + unit);
                lv[k] = *src;
            }
            asm volatile("" ::: "memory");
            RB_STAMP(1, 256);
            __syncthreads();                                   // (2) accumulators in LDS: the SV staging area is dead
            reinterpret_cast<u32x2*>(s_P)[vt] = (u32x2){0u, 0u};   // P rows of the padding heads (and everything else) = 0
            __syncthreads();                                   // (3) rotated K tile (and q) in LDS
            RB_STAMP(2, 256);
            TileAttn<AG> ta;
            ta.scores(q_lds, sOut, cnt * 8 - i0, ar.scale, vt, s_gmax);              // the host loads are still in flight
            __syncthreads();                                   // (3a) group maxima in LDS
            ta.weights(vt, s_gmax, s_P, s_gl, s_M);                                  // still in flight
            RB_STAMP(3, 256);
#pragma unroll
            for (int k = 0; k < 4; ++k) {                      // V chunks of the live slots into the cache and, every row, into
                if (lact[k]) v_buf[(long long)bh * v_stride_u128 + v_off_u128 + (long long)lslot[k] * 128 + unit] = lv[k];
                // the LDS image the MFMA reads k-major (rows of dead chunks are filler rows of the cache: weight 0)
                *reinterpret_cast<u32x4*>(s_v + skv_v_off((2 * k + rsub) * 8 + (unit >> 4), unit & 15)) = lv[k];
            }
#ifdef SKV_RB_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RB_STAMP(4, 256);
#endif
            __syncthreads();                                   // (4) V image, P, group sums in LDS
            RB_STAMP(5, 256);
            skv_tile_pv_mfma<AG>(s_v, s_P, s_gl, s_M, rec, rec_stride, tid & 63, (tid >> 6) - 4);
            RB_STAMP(6, 256);
            return;
        }
    }

    // ---- phase 1: issue every global load this workgroup needs
    // A fragments (gathered U rows)
    const int arow = i0 + wave * 16 + (lane & 15);
    long long pos = 0;
    if (arow < total_rows) {
        pos = chunk_id_at(ids, ids64, (size_t)bh * S + arow / C) * C + arow % C;
        if (pos < 0 || pos >= seq_len) pos = 0;  // invalid ids are a caller error; never fault
    }
    const bf16_t* urow = U + ((size_t)b * seq_len + pos) * R + 8 * (lane >> 4);
    u32x4 afrag[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) afrag[ks] = *reinterpret_cast<const u32x4*>(urow + 32 * ks);
    // SV[b][h] -> registers (stored to LDS below); with the attention role the staging goes in two halves (128 VGPRs instead
    // of ~170).  That was written for two 512-thread workgroups per CU; measured (38.1 vs 36.4 us, see the launcher) ONE
    // workgroup per CU is faster - the launcher's LDS request enforces it - and the half staging stayed because it costs
    // nothing there (the second half's loads are in flight while the first half is written to LDS).
    constexpr int SV_HALF = AG > 0 ? (SV_ITERS + 1) / 2 : SV_ITERS;
    u32x4 svreg[SV_HALF];
    const u32x4* const sv_src = reinterpret_cast<const u32x4*>(SV + (size_t)bh * RB_D * R);
#pragma unroll
    for (int it = 0; it < SV_HALF; ++it) {
        const int u = tid + it * 256;
        if (u < RB_D * UNITS_PER_ROW) svreg[it] = sv_src[u];
    }
    // cos / sin of the rows this thread will finish in the epilogue
    u32x4 ecos[MODE == 0 ? 1 : EPI_ITERS], esin[MODE == 0 ? 1 : EPI_ITERS];
    bool evalid[EPI_ITERS];
#pragma unroll
    for (int it = 0; it < EPI_ITERS; ++it) {
        const int u = tid + it * 256;
        const int row = u / EPI_UNITS, c = u % EPI_UNITS, i = i0 + row;
        evalid[it] = (i < total_rows) && (i >= cnt * C);
        if (MODE != 0) {   // unconditional loads (rows that are not rebuilt read table row 0): no branch, no early wait
            long long p = i < total_rows ? chunk_id_at(ids, ids64, (size_t)bh * S + i / C) * C + i % C : 0;
            if (p < 0 || p >= seq_len || !evalid[it]) p = 0;   // invalid ids are a caller error; never read outside the table
            const bf16_t* cs = cos_sin + p * cs_stride;
            if (MODE == 1) {
                ecos[it] = *reinterpret_cast<const u32x4*>(cs + 8 * c);
                esin[it] = *reinterpret_cast<const u32x4*>(cs + 64 + 8 * c);
            } else if (c < 8) {
                u32x2 c2 = *reinterpret_cast<const u32x2*>(cs + 4 * c);
                u32x2 s2 = *reinterpret_cast<const u32x2*>(cs + 32 + 4 * c);
                ecos[it][0] = c2[0]; ecos[it][1] = c2[1];
                esin[it][0] = s2[0]; esin[it][1] = s2[1];
            }
        }
    }
    int edi[EPI_ITERS];        // destination row of the epilogue rows (in-place layout), looked up before the host loads
#pragma unroll
    for (int it = 0; it < EPI_ITERS; ++it) {
        const int i = i0 + (tid + it * 256) / EPI_UNITS;
        edi[it] = (dst_slots && evalid[it]) ? dst_slots[(size_t)bh * S + i / C] * C + i % C : i;
    }
    // ---- phase 2: SV -> LDS (rows padded to 336 B: conflict-free 16-row x 16-B fragment reads)
#pragma unroll
    for (int half = 0; half * SV_HALF < SV_ITERS; ++half) {
        if (half > 0) {
#pragma unroll
            for (int it = 0; it < SV_HALF; ++it) {
                const int u = tid + (half * SV_HALF + it) * 256;
                if (half * SV_HALF + it < SV_ITERS && u < RB_D * UNITS_PER_ROW) svreg[it] = sv_src[u];
            }
        }
#pragma unroll
        for (int it = 0; it < SV_HALF; ++it) {
            const int u = tid + (half * SV_HALF + it) * 256;
            if (half * SV_HALF + it < SV_ITERS && u < RB_D * UNITS_PER_ROW) {
                const int row = u / UNITS_PER_ROW, c16 = u % UNITS_PER_ROW;
                *reinterpret_cast<u32x4*>(sSV + row * RB_SV_PITCH + c16 * 16) = svreg[it];
            }
        }
    }
    if constexpr (AG > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // see barrier (1) of the attention waves
    __syncthreads();

    // ---- phase 3: MFMA: acc[nb] covers rows wave*16.., columns nb*16..
    f32x4 acc[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 a = __builtin_bit_cast(bf16x8, afrag[ks]);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            u32x4 braw = *reinterpret_cast<const u32x4*>(sSV + (nb * 16 + (lane & 15)) * RB_SV_PITCH + ks * 64 +
                                                          (lane >> 4) * 16);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, braw), acc[nb], 0, 0, 0);
        }
    }
    // accumulators -> bf16 tile in LDS (C/D map: col = lane&15, row = (lane>>4)*4 + reg)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = wave * 16 + (lane >> 4) * 4 + r, col = nb * 16 + (lane & 15);
            *reinterpret_cast<bf16_t*>(sOut + row * RB_OUT_PITCH + col * 2) = f2bf(acc[nb][r]);
        }
    __syncthreads();

    // ---- phase 4: row-wise epilogue (RoPE in bf16 arithmetic, 16-B stores, whole 256-B rows)
#pragma unroll
    for (int it = 0; it < EPI_ITERS; ++it) {
        if (!evalid[it]) continue;
        const int u = tid + it * 256;
        const int row = u / EPI_UNITS, c = u % EPI_UNITS;
        const int di = edi[it];
        bf16_t* orow = out + (size_t)b * out_stride_b + (size_t)h * out_stride_h + (size_t)(out_row0 + di) * out_stride_s;
        if (MODE == 0) {
            *reinterpret_cast<u32x4*>(orow + 8 * c) = *reinterpret_cast<const u32x4*>(sOut + row * RB_OUT_PITCH + c * 16);
        } else if (MODE == 1) {
            // NeoX half split: unit c pairs elements [8c, 8c+8) with [64+8c, 64+8c+8)
            u32x4 x1 = *reinterpret_cast<const u32x4*>(sOut + row * RB_OUT_PITCH + c * 16);
            u32x4 x2 = *reinterpret_cast<const u32x4*>(sOut + row * RB_OUT_PITCH + 128 + c * 16);
            u32x4 o1, o2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a0 = bf_lo(x1[j]), a1 = bf_hi(x1[j]), b0 = bf_lo(x2[j]), b1 = bf_hi(x2[j]);
                float c0 = bf_lo(ecos[it][j]), c1 = bf_hi(ecos[it][j]), s0 = bf_lo(esin[it][j]), s1 = bf_hi(esin[it][j]);
                o1[j] = pack_bf2(bfr(a0 * c0) + bfr(-b0 * s0), bfr(a1 * c1) + bfr(-b1 * s1));
                o2[j] = pack_bf2(bfr(b0 * c0) + bfr(a0 * s0), bfr(b1 * c1) + bfr(a1 * s1));
            }
            *reinterpret_cast<u32x4*>(orow + 8 * c) = o1;
            *reinterpret_cast<u32x4*>(orow + 64 + 8 * c) = o2;
            if constexpr (AG > 0) {   // the tile attention reads the rotated row (each unit rewrites only what it read)
                *reinterpret_cast<u32x4*>(sOut + row * RB_OUT_PITCH + c * 16) = o1;
                *reinterpret_cast<u32x4*>(sOut + row * RB_OUT_PITCH + 128 + c * 16) = o2;
            }
        } else {
            // GLM: dims 0..63 interleaved pairs (2t, 2t+1) with cos = cs[t], sin = cs[32+t]; 64..127 copied
            u32x4 x = *reinterpret_cast<const u32x4*>(sOut + row * RB_OUT_PITCH + c * 16);
            u32x4 o = x;
            if (c < 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xe = bf_lo(x[j]), xo = bf_hi(x[j]);
                    uint32_t cw = ecos[it][j >> 1], sw = esin[it][j >> 1];
                    float cv = (j & 1) ? bf_hi(cw) : bf_lo(cw);
                    float sv = (j & 1) ? bf_hi(sw) : bf_lo(sw);
                    o[j] = pack_bf2(bfr(xe * cv) + bfr(-xo * sv), bfr(xo * cv) + bfr(xe * sv));
                }
            }
            *reinterpret_cast<u32x4*>(orow + 8 * c) = o;
            if constexpr (AG > 0) *reinterpret_cast<u32x4*>(sOut + row * RB_OUT_PITCH + c * 16) = o;
        }
    }
    if constexpr (AG > 0) {
        __syncthreads();                                       // (3) rotated K rows of the tile are in LDS
        __syncthreads();                                       // (3a) the landing waves' group maxima
        __syncthreads();                                       // (4) their V image and weights: they finish the tile alone
    }
}

#include "skv_launch.h"

// mode: 0 pre-RoPE (legacy batch_gather_gemm), 1 Llama, 2 GLM
int skv_launch_rebuild(const void* U, const void* SV, const void* cos_sin, const void* ids, int ids64,
                       const int32_t* cnts, void* out, int bs, int heads, int seq_len, int head_dim, int R, int S,
                       int C, long long cs_stride, long long out_stride_b, long long out_stride_h,
                       long long out_stride_s, int out_row0, int mode, const void* hit_temp, const int32_t* hit_offsets,
                       const int32_t* dst_slots, const void* v_host, void* v_buf, const void* v_temp, long long v_host_stride,
                       long long v_stride, long long v_off, hipStream_t st, const AttnLaunch* attn, const EarlyConsume* early) {
    if (head_dim != RB_D || C < 1 || S < 1) return SKV_ERR_UNSUPPORTED;
    if (R != 160 && R != 128 && R != 96 && R != 64) return SKV_ERR_UNSUPPORTED;  // instantiated ranks (LDS pitch fits <= 160)
    if ((out_stride_s % 8) || (out_stride_h % 8) || (out_stride_b % 8)) return SKV_ERR_ARG;
    if (mode == 1 && (cs_stride < 128 || cs_stride % 8)) return SKV_ERR_ARG;
    if (mode == 2 && (cs_stride < 64 || cs_stride % 4)) return SKV_ERR_ARG;
    if (mode < 0 || mode > 2) return SKV_ERR_ARG;
    if (hit_temp && (C != 8 || out_stride_s != 128 || !hit_offsets)) return SKV_ERR_UNSUPPORTED;
    const int tiles = (S * C + RB_ROWS - 1) / RB_ROWS;
    const size_t smem = RB_D * RB_SV_PITCH + RB_ROWS * RB_OUT_PITCH;
    int land_blocks = 0;
    if (v_buf) {
        if (!v_host || (!v_temp && !dst_slots) || !hit_offsets || (v_host_stride % 8) || (v_stride % 8) || (v_off % 8)) return SKV_ERR_ARG;
        land_blocks = attn ? 0 : (S + 7) / 8;     // attention role: the rebuild tiles land their own V chunks
    }
    AttnRole ar{};
    if (early && !attn) {            // (with the attention role the early fields arrive in AttnLaunch)
        if (!early->early_of || !early->early_staging || early->early_chunks < 1 || early->early_max < 1 || !v_buf) return SKV_ERR_ARG;
        ar.early_of = early->early_of;
        ar.early_staging = (const u32x4*)early->early_staging;
        ar.early_chunks = early->early_chunks;
        ar.early_max = early->early_max;
    }
    int attn_g = 0;
    size_t smem_all = smem;
    if (attn) {
        // attention role: in-place layout, rank 160, G in {4, 8}, V buffer laid out like the K buffer
        if (!dst_slots || !v_buf || mode == 0 || R != 160 || (attn->G != 4 && attn->G != 8) || attn->splits < 1 ||
            C != 8 || S % 8 || attn->rec_splits != attn->splits + tiles || v_stride != out_stride_h || out_stride_s != RB_D ||
            out_stride_b != (long long)heads * out_stride_h || v_off != (long long)out_row0 * RB_D)
            return SKV_ERR_UNSUPPORTED;
        if (attn->resident_sets < S || attn->kv_rows < out_row0 + attn->resident_sets * C ||
            (long long)attn->kv_rows * RB_D > out_stride_h ||
            (!attn->kv_len_dev && (attn->kv_len_host < 1 || attn->kv_len_host > attn->kv_rows)))
            return SKV_ERR_ARG;
        if (attn->early_of && (!attn->early_staging || attn->early_chunks < 1 || attn->early_max < 1)) return SKV_ERR_ARG;
        ar = AttnRole{(const bf16_t*)attn->q, (float*)attn->ws, attn->kv_len_dev, attn->kv_len_host, attn->kv_rows,
                      attn->splits, attn->rec_splits, attn->resident_sets * C, attn->scale, attn->early_of,
                      (const u32x4*)attn->early_staging, attn->early_chunks, attn->early_max};
        attn_g = attn->G;
        // fused tile: SV staging (later: the tile attention's V image / weights) | K tile | q
        const size_t part = (size_t)AT_GROUPS * attn_g * (AT_D + 2) * sizeof(float);
        smem_all = (size_t)RB_D * RB_SV_PITCH + (size_t)RB_ROWS * RB_OUT_PITCH + (size_t)attn_g * 256 /* q */;
        const size_t need = part + (size_t)S * sizeof(int);    // resident-rows role: group partials + the slot list
        if (need > smem_all) smem_all = need;
        // ONE workgroup per CU (LDS request above half of the 160 KB): measured with in-kernel stamps and A/B runs, a CU
        // with host-memory loads outstanding serves its other memory traffic only when they return, so a second workgroup
        // sharing the CU (rebuild or attention role) stalls behind the first one's PCIe round trip (bs 1: 38.1 vs 36.4 us)
        if (smem_all < 84 * 1024) smem_all = 84 * 1024;
    }
    dim3 grid(tiles + land_blocks, bs * heads), block(256);
    if (attn) {
        grid = dim3((tiles + attn->splits) * bs * heads, 1);
        block = dim3(512);
    }
    if (attn) {
#define SKV_RBA(M, GG)                                                                                             \
    do {                                                                                                           \
        static size_t attr_bytes[64] = {};                                                                         \
        if (skv_ensure_max_lds((const void*)skv_rebuild_kernel<M, 5, GG>, smem_all, attr_bytes) != SKV_OK)         \
            return SKV_ERR_LAUNCH;                                                                                 \
        hipLaunchKernelGGL((skv_rebuild_kernel<M, 5, GG>), grid, block, smem_all, st, (const bf16_t*)U,            \
                           (const bf16_t*)SV, (const bf16_t*)cos_sin, ids, cnts, (bf16_t*)out, heads, seq_len, S, C, \
                           ids64, cs_stride, out_stride_b, out_stride_h, out_stride_s, out_row0,                   \
                           (const bf16_t*)hit_temp, hit_offsets, dst_slots, tiles, (const u32x4*)v_host,           \
                           (u32x4*)v_buf, (const u32x4*)v_temp, v_host_stride / 8, v_stride / 8, v_off / 8,        \
                           land_blocks, ar);                                                                       \
    } while (0)
        if (mode == 1) { if (attn_g == 4) SKV_RBA(1, 4); else SKV_RBA(1, 8); }
        else           { if (attn_g == 4) SKV_RBA(2, 4); else SKV_RBA(2, 8); }
#undef SKV_RBA
        return SKV_OK;
    }
#define SKV_RB(M, K)                                                                                               \
    hipLaunchKernelGGL((skv_rebuild_kernel<M, K>), grid, block, smem, st, (const bf16_t*)U, (const bf16_t*)SV,     \
                       (const bf16_t*)cos_sin, ids, cnts, (bf16_t*)out, heads, seq_len, S, C, ids64, cs_stride,    \
                       out_stride_b, out_stride_h, out_stride_s, out_row0, (const bf16_t*)hit_temp, hit_offsets, dst_slots, tiles, \
                       (const u32x4*)v_host, (u32x4*)v_buf, (const u32x4*)v_temp, v_host_stride / 8, v_stride / 8, v_off / 8, \
                       land_blocks, ar)
#define SKV_RB_K(M)                 \
    switch (R / 32) {               \
        case 5: SKV_RB(M, 5); break; \
        case 4: SKV_RB(M, 4); break; \
        case 3: SKV_RB(M, 3); break; \
        default: SKV_RB(M, 2); break; \
    }
    if (mode == 0) { SKV_RB_K(0) } else if (mode == 1) { SKV_RB_K(1) } else { SKV_RB_K(2) }
#undef SKV_RB_K
#undef SKV_RB
    return SKV_OK;
}
