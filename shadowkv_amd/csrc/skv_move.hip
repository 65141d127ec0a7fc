// Chunk-row movement inside the sparse region.
//
// After the diff (skv_select.hip) slot i of the sparse region must hold chunk
// cached_pos_ids[i]:   i <  cnt : row i <- row offsets[i]   (hit, already resident, offsets[i] >= i)
//                      i >= cnt : row i <- host_table[offsets[i]]   (miss; V only)
// One row = one 8-token chunk = 2 KiB = 128 lanes x 16 B.
//
// Replaces /root/reference/kernels/copy.cuh:785-846 (gather_copy_var_midpoint_BP: 2 CTAs per
// (batch, head), 32-row shared-memory staging, `temp` bounce buffer, flag hand-off between the two CTAs)
// and copy.cuh:649-687 (gather_copy_d2d).  MI355X design: NO in-kernel synchronisation between workgroups.
// A (batch, head) is served by S/8 workgroups of two launches on one stream:
//   phase 1  skv_stage_hits_kernel : temp[b][i] <- buf[b][sparse + offsets[i]]   for hit rows that move
//   phase 2  skv_land_rows_kernel  : buf[b][sparse + i] <- temp[b][i] (moved hits), <- host[offsets[i]] (misses)
// The kernel boundary orders "every read of the old layout" before "any write of the new one"; host rows come
// over PCIe as plain 16-B loads from mapped pinned memory.  Rows that do not move (offsets[i] == i) are neither
// read nor written.  The reference's `temp` argument is the staging buffer (same size: [bs][heads][S][1024] bf16).
// (Round 1 also had a single-launch variant with a per-team arrival counter; it needed every workgroup of a team
// co-resident and could only flag a timeout after the fact - removed: nothing spins in this library.)
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"

#define SKV_ROWS_PER_WG 8
#define SKV_ROW_U128 128  // 16-byte units per chunk row

// no-cache gather: row i <- host_table[position_ids[i]] for every i (int64 ids)
// (/root/reference/kernels/copy.cuh:481-508 gahter_copy_fixed_start_end)
__global__ __launch_bounds__(256) void skv_gather_rows_kernel(const u32x4* __restrict__ host_rows,
                                                              u32x4* __restrict__ dev,
                                                              const int64_t* __restrict__ ids,
                                                              long long host_stride_u128,
                                                              long long dev_stride_u128, int S) {
    const int b = blockIdx.y;
    const int tid = threadIdx.x, unit = tid & 127, rsub = tid >> 7;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * SKV_ROWS_PER_WG + k * 2 + rsub;
        if (i < S) {
            const long long src = ids[(size_t)b * S + i];
            dev[(long long)b * dev_stride_u128 + (long long)i * SKV_ROW_U128 + unit] =
                host_rows[(long long)b * host_stride_u128 + src * SKV_ROW_U128 + unit];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Two-phase movement used by the fused decode path (no in-kernel synchronisation at all):
//   phase 1  skv_stage_hits_kernel : temp[b][i] <- buf[b][sparse + offsets[i]]   for hit rows that move
//   phase 2  skv_land_rows_kernel  : buf[b][sparse + i] <- temp[b][i] (moved hits), <- host[offsets[i]] (misses)
// The kernel boundary orders "every read of the old layout" before "any write of the new one".  A single-kernel
// variant with an inter-workgroup counter (round 1) was removed: nothing in this library spins on another workgroup,
// and while a V launch pulls its miss rows over PCIe, agent-scope atomics / polls of a concurrent K compaction were
// served only when the PCIe reads drained (measured 37 us instead of 15 us alone).
// Costs one extra HBM round trip for the moved hit rows (<= 2 x 4 MB per layer), saves every spin.
// grid (ceil(S/8), B, nbuf): z selects the buffer (0 = K, 1 = V).
// ---------------------------------------------------------------------------------------------------
struct MoveBuf {
    u32x4* buf;      // cache buffer [B][stride]
    u32x4* temp;     // [B][S][128]
};

__global__ __launch_bounds__(256) void skv_stage_hits_kernel(MoveBuf b0, MoveBuf b1, const int32_t* __restrict__ offsets,
                                                             const int32_t* __restrict__ cnts, long long stride_u128,
                                                             long long off_u128, int S) {
    const MoveBuf mb = blockIdx.z == 0 ? b0 : b1;
    const int b = blockIdx.y, tid = threadIdx.x, unit = tid & 127, rsub = tid >> 7;
    const int cnt = cnts[b];
    if (blockIdx.x * SKV_ROWS_PER_WG >= cnt) return;
    u32x4 v[4];
    bool act[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * SKV_ROWS_PER_WG + k * 2 + rsub;
        act[k] = false;
        if (i < cnt) {
            const int off = offsets[(size_t)b * S + i];
            if (off != i) {
                act[k] = true;
                v[k] = mb.buf[(long long)b * stride_u128 + off_u128 + (long long)off * SKV_ROW_U128 + unit];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * SKV_ROWS_PER_WG + k * 2 + rsub;
        if (act[k]) mb.temp[((long long)b * S + i) * SKV_ROW_U128 + unit] = v[k];
    }
}

__global__ __launch_bounds__(256) void skv_land_rows_kernel(const u32x4* __restrict__ host_rows, u32x4* __restrict__ buf,
                                                            const u32x4* __restrict__ temp,
                                                            const int32_t* __restrict__ offsets,
                                                            const int32_t* __restrict__ cnts, long long host_stride_u128,
                                                            long long stride_u128, long long off_u128, int S) {
    const int b = blockIdx.y, tid = threadIdx.x, unit = tid & 127, rsub = tid >> 7;
    const int cnt = cnts[b];
    u32x4 v[4];
    bool act[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * SKV_ROWS_PER_WG + k * 2 + rsub;
        act[k] = false;
        if (i < S) {
            const int off = offsets[(size_t)b * S + i];
            if (i < cnt) {
                if (off != i) {
                    act[k] = true;
                    v[k] = temp[((long long)b * S + i) * SKV_ROW_U128 + unit];
                }
            } else if (host_rows != nullptr) {
                act[k] = true;
                v[k] = host_rows[(long long)b * host_stride_u128 + (long long)off * SKV_ROW_U128 + unit];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * SKV_ROWS_PER_WG + k * 2 + rsub;
        if (act[k]) buf[(long long)b * stride_u128 + off_u128 + (long long)i * SKV_ROW_U128 + unit] = v[k];
    }
}

int skv_launch_stage_hits(void* k_buf, void* k_temp, void* v_buf, void* v_temp, const int32_t* offsets,
                          const int32_t* cnts, long long stride_elems, long long off_elems, int B, int S,
                          hipStream_t st) {
    if (S < 1 || B < 1 || (stride_elems % 8) || (off_elems % 8)) return SKV_ERR_ARG;
    MoveBuf b0{(u32x4*)k_buf, (u32x4*)k_temp}, b1{(u32x4*)v_buf, (u32x4*)v_temp};
    int nbuf = 2;
    if (!k_buf) { b0 = b1; nbuf = 1; }
    else if (!v_buf) nbuf = 1;
    hipLaunchKernelGGL(skv_stage_hits_kernel, dim3((S + SKV_ROWS_PER_WG - 1) / SKV_ROWS_PER_WG, B, nbuf), dim3(256), 0, st,
                       b0, b1, offsets, cnts, stride_elems / 8, off_elems / 8, S);
    return SKV_OK;
}

int skv_launch_land_rows(const void* host_rows, void* buf, const void* temp, const int32_t* offsets,
                         const int32_t* cnts, long long host_len_elems, long long stride_elems, long long off_elems,
                         int B, int S, hipStream_t st) {
    if (S < 1 || B < 1 || (host_len_elems % 8) || (stride_elems % 8) || (off_elems % 8)) return SKV_ERR_ARG;
    hipLaunchKernelGGL(skv_land_rows_kernel, dim3((S + SKV_ROWS_PER_WG - 1) / SKV_ROWS_PER_WG, B), dim3(256), 0, st,
                       (const u32x4*)host_rows, (u32x4*)buf, (const u32x4*)temp, offsets, cnts, host_len_elems / 8,
                       stride_elems / 8, off_elems / 8, S);
    return SKV_OK;
}

// Legacy entry points (gather_copy_with_offsets / gather_copy_d2d_with_offsets): stage + land, two launches.
// lengths / offsets / strides are in bf16 elements (reference convention,
// /root/reference/models/kv_cache.py:1090-1093); one row = 1024 elements.  temp: [B][S][1024] bf16.
int skv_launch_move_rows(const void* host_rows, void* dev, void* temp, const int32_t* offsets, const int32_t* cnts,
                         long long host_len_elems, long long dev_stride_elems, long long dev_off_elems, int B, int S,
                         hipStream_t st) {
    if (!temp) return SKV_ERR_ARG;
    int rc = skv_launch_stage_hits(nullptr, nullptr, dev, temp, offsets, cnts, dev_stride_elems, dev_off_elems, B, S, st);
    if (rc != SKV_OK) return rc;
    return skv_launch_land_rows(host_rows, dev, temp, offsets, cnts, host_len_elems, dev_stride_elems, dev_off_elems, B,
                                S, st);
}

int skv_launch_gather_rows(const void* host_rows, void* dev, const int64_t* ids, long long host_len_elems,
                           long long dev_len_elems, int B, int S, hipStream_t st) {
    if (S < 1 || B < 1 || (host_len_elems % 8) || (dev_len_elems % 8)) return SKV_ERR_ARG;
    hipLaunchKernelGGL(skv_gather_rows_kernel, dim3((S + SKV_ROWS_PER_WG - 1) / SKV_ROWS_PER_WG, B), dim3(256), 0,
                       st, (const u32x4*)host_rows, (u32x4*)dev, ids, host_len_elems / 8, dev_len_elems / 8, S);
    return SKV_OK;
}
