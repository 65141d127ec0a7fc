// Chunk-row movement inside the sparse region.
//
// After the diff (skv_select.hip) slot i of the sparse region must hold chunk
// cached_pos_ids[i]:   i <  cnt : row i <- row offsets[i]   (hit, already resident, offsets[i] >= i)
//                      i >= cnt : row i <- host_table[offsets[i]]   (miss; V only)
// One row = one 8-token chunk = 2 KiB = 128 lanes x 16 B.
//
// Replaces /root/reference/kernels/copy.cuh:785-846 (gather_copy_var_midpoint_BP: 2 CTAs per
// (batch, head), 32-row shared-memory staging, `temp` bounce buffer, flag hand-off) and
// copy.cuh:649-687 (gather_copy_d2d).  MI355X design: a (batch, head) is served by a TEAM of
// S/8 workgroups (256 teams*WGs at the headline config instead of 16 CTAs); every workgroup
// first LOADS its 8 destination rows' sources into registers (16 KiB per workgroup in flight,
// host rows come over PCIe as plain 16-B loads from mapped pinned memory), bumps the team
// counter, waits until the whole team has finished reading, then STORES.  In-place movement is
// therefore race-free without a bounce buffer: nothing is written before everything was read.
// Rows that do not move (offsets[i] == i) are neither read nor written.
//
// The counter is only an ordering device (no data is handed between workgroups through
// memory), so a relaxed agent-scope atomic + drained loads is sufficient; every spin is
// bounded.  `signals[b]` is zero on entry and zero on exit, like the reference's.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"

#define SKV_ROWS_PER_WG 8
#define SKV_ROW_U128 128  // 16-byte units per chunk row

__device__ unsigned int g_skv_move_timeout = 0;

__global__ __launch_bounds__(256) void skv_move_rows_kernel(
    const u32x4* __restrict__ host_rows,  // pinned host table (nullable: no host part)
    u32x4* dev,                           // device buffer
    const int32_t* __restrict__ offsets,  // [B][S]
    const int32_t* __restrict__ cnts,     // [B]
    unsigned int* signals,                // [B], zero on entry / exit
    long long host_stride_u128,           // per-(batch,head) stride of the host table, 16-B units
    long long dev_stride_u128,            // per-(batch,head) stride of the device buffer
    long long dev_off_u128,               // offset of the sparse region inside a (batch,head) block
    int S, int wgs_per_team, int B, int teams) {
    const int team = blockIdx.x / wgs_per_team, w = blockIdx.x % wgs_per_team;
    const int tid = threadIdx.x, unit = tid & 127, rsub = tid >> 7;
    for (int b = team; b < B; b += teams) {
        const int cnt = cnts[b];
        u32x4 v[4];
        bool act[4];
        u32x4* dbase = dev + (long long)b * dev_stride_u128 + dev_off_u128;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = w * SKV_ROWS_PER_WG + k * 2 + rsub;
            act[k] = false;
            if (i < S) {
                const int off = offsets[(size_t)b * S + i];
                if (i < cnt) {
                    if (off != i) {
                        act[k] = true;
                        v[k] = dbase[(long long)off * SKV_ROW_U128 + unit];
                    }
                } else if (host_rows != nullptr) {
                    act[k] = true;
                    v[k] = host_rows[(long long)b * host_stride_u128 + (long long)off * SKV_ROW_U128 + unit];
                }
            }
        }
        // every source byte is in registers before this workgroup reports "read done"
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(&signals[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Poll with backoff: a team waits for its slowest member (PCIe rows: tens of us), and every
            // poll is an uncached access to one line that the arriving atomics also need - hundreds of
            // workgroups polling every 0.1 us saturate that L2 channel and delay the arrivals themselves.
            unsigned spins = 0;
            while ((__hip_atomic_load(&signals[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu) <
                   (unsigned)wgs_per_team) {
                if (spins < 4) __builtin_amdgcn_s_sleep(8);
                else if (spins < 12) __builtin_amdgcn_s_sleep(24);
                else __builtin_amdgcn_s_sleep(48);
                if (++spins > (1u << 22)) {
                    atomicOr(&g_skv_move_timeout, 1u);
                    break;
                }
            }
        }
        __syncthreads();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = w * SKV_ROWS_PER_WG + k * 2 + rsub;
            if (act[k]) dbase[(long long)i * SKV_ROW_U128 + unit] = v[k];
        }
        if (tid == 0) {
            unsigned old = __hip_atomic_fetch_add(&signals[b], 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old >> 16) == (unsigned)(wgs_per_team - 1))
                __hip_atomic_store(&signals[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

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
// The kernel boundary orders "every read of the old layout" before "any write of the new one".  Measured
// reason for not using the single-kernel team counter here: while the V launch is pulling its miss rows
// over PCIe, agent-scope atomics / sc1 polls of a concurrently running K compaction are served only when
// the PCIe reads drain (the K launch always ended with the V launch, 37 us instead of 15 us alone).
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

// lengths / offsets / strides are in bf16 elements (reference convention,
// /root/reference/models/kv_cache.py:1090-1093); one row = 1024 elements.
int skv_launch_move_rows(const void* host_rows, void* dev, const int32_t* offsets, const int32_t* cnts,
                         unsigned int* signals, long long host_len_elems, long long dev_stride_elems,
                         long long dev_off_elems, int B, int S, hipStream_t st) {
    if (S < 1 || B < 1) return SKV_ERR_ARG;
    if ((host_len_elems % 8) || (dev_stride_elems % 8) || (dev_off_elems % 8)) return SKV_ERR_ARG;
    const int wgs_per_team = (S + SKV_ROWS_PER_WG - 1) / SKV_ROWS_PER_WG;
    if (wgs_per_team > 1024) return SKV_ERR_UNSUPPORTED;
    int teams = 1024 / wgs_per_team;  // keep the whole grid resident: every team spins on its members
    if (teams > B) teams = B;
    hipLaunchKernelGGL(skv_move_rows_kernel, dim3(teams * wgs_per_team), dim3(256), 0, st, (const u32x4*)host_rows,
                       (u32x4*)dev, offsets, cnts, signals, host_len_elems / 8, dev_stride_elems / 8,
                       dev_off_elems / 8, S, wgs_per_team, B, teams);
    return SKV_OK;
}

int skv_launch_gather_rows(const void* host_rows, void* dev, const int64_t* ids, long long host_len_elems,
                           long long dev_len_elems, int B, int S, hipStream_t st) {
    if (S < 1 || B < 1 || (host_len_elems % 8) || (dev_len_elems % 8)) return SKV_ERR_ARG;
    hipLaunchKernelGGL(skv_gather_rows_kernel, dim3((S + SKV_ROWS_PER_WG - 1) / SKV_ROWS_PER_WG, B), dim3(256), 0,
                       st, (const u32x4*)host_rows, (u32x4*)dev, ids, host_len_elems / 8, dev_len_elems / 8, S);
    return SKV_OK;
}

extern "C" int skv_move_timeout_flag(void) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_skv_move_timeout), sizeof(v)) != hipSuccess) return -1;
    return (int)v;
}
