// REJECTED EXPERIMENT - not part of libshadowkv_hip.so (profiles/r03_merge_oproj_fused.txt).  It was built as shadowkv_amd/csrc/skv_merge_oproj.hip
// (C entry skv_attn_finish_oproj_inplace; the merge body it calls now lives in skv_attn.hip again).
// Merge of the attention records + O projection (one token, one sequence) as ONE launch (round 3).
//
// As two launches the merge costs 5.4 us (two thirds of it launch boundary and one memory round trip: 32 workgroups, 1.4 MB)
// and the O projection 7.3 us (33.5 MB at 6.4 TB/s = 5.3 us + 1.9 us around the stream): the merge leaves HBM idle, the
// projection cannot start before it.  Here the first Hq workgroups of the grid merge one query head each (the body of
// skv_attn_merge_kernel), every other workgroup is a GEMV workgroup (the arithmetic of skv_gemv_kernel<2, ...>: 4 waves x 2
// rows, lane l owns elements 8l..8l+7 of each 512-element step, k ascending, same final tree) that requests ALL its
// weights first - 16 KiB per wave, they stream in while the merge runs - and only then waits for the merged vector:
// the merge workgroups publish with a release fence + one add to an arrival counter, the GEMV workgroups poll it (one wave
// per workgroup, bounded), acquire, read x (8 KiB) and finish from registers.  Outputs are bit-identical to the two
// launches.  Unlike the fetch launch (profiles/r03_in_launch_merge.txt) nothing reads host memory here, so the agent-scope
// release costs < 1 us; unlike the whole dense tail (profiles/r03_layer_tail_megakernel.txt) the dependent chain stands in
// front of an IDLE memory system: the merge's record loads are issued with the first weight requests, not behind 48 MB.
//
// Co-residency: the GEMV workgroups spin, so every merge workgroup must be running or able to start - the launcher refuses
// the launch unless the WHOLE grid is resident at once (occupancy query), which holds with room to spare (544 workgroups,
// > 1,000 slots); the poll is bounded all the same and reports through skv_merge_oproj_status.
// Launch bookkeeping without a reset: `ticket` counts every workgroup that has finished polling (or merging) - exactly
// `grid` per launch, so ticket / grid read at workgroup entry is the launch index k whatever the other workgroups of this
// launch have done so far - and `arrived` counts merge workgroups: the GEMV workgroups wait for arrived >= (k + 1) * Hq.
#include "../../include/shadowkv_hip.h"
#include "skv_attn_body.h"
#include "skv_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Phase stamps (diagnostic build only, -DSKV_RB_STAMPS -> libshadowkv_hip_stamps.so; tools/merge_oproj_probe.py): 8 words per
// workgroup: merge role 0 start, 1 merged, 2 arrived; GEMV role 0 start, 3 weights requested, 4 poll done, 5 x here, 6 end.
#ifdef SKV_RB_STAMPS
__device__ unsigned long long g_mo_stamps[1024 * 8];
#define MO_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_mo_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
extern "C" __attribute__((visibility("default"))) int skv_debug_mo_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mo_stamps), sizeof(g_mo_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define MO_STAMP(i)
#endif

#define MO_POLL_LIMIT (1 << 16)
#define MO_K 4096                       // K of the projection = q_heads * 128
#define MO_KSTEPS (MO_K / 512)

__global__ __launch_bounds__(256) void skv_merge_oproj_kernel(const float* __restrict__ ws, const int32_t* __restrict__ cnts,
                                                              bf16_t* attn_out /* [Hq * 128], written by the merge role */,
                                                              int G, int splits, int tiles, int n_merge,
                                                              const bf16_t* __restrict__ W /* [N][4096] */, bf16_t* __restrict__ y,
                                                              int N, unsigned long long* sync /* ticket, arrived, status */) {
    __shared__ __attribute__((aligned(16))) float s_rec[MRG_MAX_REC * AT_REC];
    __shared__ float s_wgt[MRG_MAX_REC];
    __shared__ float s_a[2][AT_D];
    __shared__ float s_l[2];
    __shared__ unsigned long long s_k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* const ticket = sync;
    unsigned long long* const arrived = sync + 16;           // (own 128-B line)
    unsigned long long tk = 0;                                // (requested first, consumed behind the weight requests)
    if (tid == 0) tk = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    MO_STAMP(0);
    if ((int)blockIdx.x < n_merge) {
        // ---- merge role: query head blockIdx.x
        skv_merge_head_body(ws, cnts, attn_out, G, splits, tiles, blockIdx.x, tid, s_rec, s_wgt, s_a, s_l);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of the merged vector has reached L2
        __syncthreads();                                     // ... and every other wave's
        MO_STAMP(1);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // L2 write-back: visible to the other XCDs
            __hip_atomic_fetch_add(arrived, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef SKV_RB_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MO_STAMP(2);
#endif
        return;
    }
    // ---- GEMV role: rows (block * 4 + wave) * 2 + {0, 1}
    const int unit0 = (((int)blockIdx.x - n_merge) * 4 + wave) * 2;
    const bf16_t* wp[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) wp[r] = W + (size_t)min(unit0 + r, N - 1) * MO_K + 8 * lane;   // (clamped rows: computed, dropped)
#ifndef SKV_MO_HEAD_START
#define SKV_MO_HEAD_START 40                                  // x 64 clocks (~1 us): see below
#endif
    // the merge workgroups' record loads go first: a memory system that already holds this launch's 33.5 MB of weight
    // requests serves them 3 us later (in-kernel stamps: merged at 6.2 us instead of ~3)
    if (SKV_MO_HEAD_START > 0) __builtin_amdgcn_s_sleep(SKV_MO_HEAD_START);
    u32x4 wv[2][MO_KSTEPS];
#pragma unroll
    for (int u = 0; u < MO_KSTEPS; ++u)
#pragma unroll
        for (int r = 0; r < 2; ++r) wv[r][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + (size_t)u * 512));
    MO_STAMP(3);
    if (tid == 0) s_k = tk / gridDim.x;                      // launch index (see the header)
    __syncthreads();
    if (wave == 0) {
        const unsigned long long target = (s_k + 1ull) * (unsigned long long)n_merge;
        int polls = 0;
        for (;;) {
            unsigned long long c = 0;
            if (lane == 0) c = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(c & 0xffffffffull)) |
                ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(c >> 32)) << 32);
            if (c >= target) break;
            if (++polls > MO_POLL_LIMIT) {                   // cannot happen with the whole grid resident
                if (lane == 0) atomicExch(reinterpret_cast<unsigned int*>(sync + 32), 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // x: the merged vector, read ONCE per workgroup with agent-coherent (sc1) loads into LDS - no L2 invalidate (a
        // buffer_inv per wave made the 2,048 waves' reads of the same 8 KiB miss L2 again and again: 8-18 us), and 4 MB of
        // reads of those few lines instead of 16
        {
            const unsigned int* xs = reinterpret_cast<const unsigned int*>(attn_out);
            unsigned int t[MO_K / 2 / 64];
#pragma unroll
            for (int i = 0; i < MO_K / 2 / 64; ++i)
                t[i] = __hip_atomic_load(xs + i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < MO_K / 2 / 64; ++i) reinterpret_cast<unsigned int*>(s_rec)[i * 64 + lane] = t[i];
        }
    }
    MO_STAMP(4);
    __syncthreads();
    u32x4 xv[MO_KSTEPS];
#pragma unroll
    for (int u = 0; u < MO_KSTEPS; ++u) xv[u] = reinterpret_cast<const u32x4*>(s_rec)[u * 64 + lane];
#ifdef SKV_RB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MO_STAMP(5);
#endif
    f32x2 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MO_KSTEPS; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 xx = (f32x2){bf_lo(xv[u][j]), bf_hi(xv[u][j])};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const f32x2 ww = (f32x2){bf_lo(wv[r][u][j]), bf_hi(wv[r][u][j])};
                acc[r][j] = __builtin_elementwise_fma(ww, xx, acc[r][j]);
            }
        }
    float tot[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float s = ((acc[r][0].x + acc[r][0].y) + (acc[r][1].x + acc[r][1].y)) +
                        ((acc[r][2].x + acc[r][2].y) + (acc[r][3].x + acc[r][3].y));
        tot[r] = wave_tree_sum(s);
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (unit0 + r < N) y[unit0 + r] = f2bf(tot[r]);
    }
    MO_STAMP(6);
}

extern "C" size_t skv_merge_oproj_workspace_bytes(void) { return 48 * sizeof(unsigned long long); }

extern "C" int skv_attn_finish_oproj_inplace(const void* attn_workspace, const int32_t* cnts, void* attn_out, int q_heads,
                                             int kv_heads, int select_sets, int attn_splits, const void* Wo, void* o_out,
                                             int hidden, void* sync_workspace, skv_stream_t stream) {
    if (!attn_workspace || !cnts || !attn_out || !Wo || !o_out || !sync_workspace) return SKV_ERR_ARG;
    if (kv_heads < 1 || q_heads % kv_heads || select_sets < 8 || select_sets % 8 || attn_splits < 1 || hidden < 1) return SKV_ERR_ARG;
    if (attn_splits + select_sets / 8 > MRG_MAX_REC || q_heads * AT_D != MO_K) return SKV_ERR_UNSUPPORTED;
    const int grid = q_heads + (hidden + 7) / 8;
    static int slots[64];                                     // workgroups of this kernel the device holds at once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SKV_ERR_LAUNCH;
    if (!slots[dev]) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, skv_merge_oproj_kernel, 256, 0) != hipSuccess)
            return SKV_ERR_LAUNCH;
        slots[dev] = cus * per_cu > 0 ? cus * per_cu : -1;
    }
    if (slots[dev] < grid) return SKV_ERR_UNSUPPORTED;        // the GEMV workgroups spin: the whole grid must be resident
    hipLaunchKernelGGL(skv_merge_oproj_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)attn_workspace, cnts,
                       (bf16_t*)attn_out, q_heads / kv_heads, attn_splits, select_sets / 8, q_heads, (const bf16_t*)Wo,
                       (bf16_t*)o_out, hidden, (unsigned long long*)sync_workspace);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

extern "C" int skv_merge_oproj_status(const void* sync_workspace, int* status_out) {
    if (!sync_workspace || !status_out) return SKV_ERR_ARG;
    unsigned long long v = 0;
    if (hipMemcpy(&v, (const unsigned long long*)sync_workspace + 32, 8, hipMemcpyDeviceToHost) != hipSuccess) return SKV_ERR_LAUNCH;
    *status_out = (int)(v & 0xffffffffull);
    return SKV_OK;
}
