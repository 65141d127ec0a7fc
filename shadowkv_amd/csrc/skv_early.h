// Speculative early V fetch (round 3; VERDICT r2 item 4 "shorten the 22 us in front of the link") - device roles.
//
// The V rows of a step's miss chunks can only be requested over PCIe once the top-k has named them, i.e. behind scan +
// normalise + top-k (8 + 5 + 9 us at the headline shape, 10 + 7 + 12 at GLM-4 200K), and the link is what bounds the fetch
// launch (1.4 MB at 55 GB/s = 26 us).  Which chunks will miss is predictable right after the SCAN: the k-th largest score
// moves very little from step to step, so "this slot's logit reaches last step's threshold, and its chunk is not resident"
// names 98 % of the step's misses with 8 % extra (tools/spec_fetch_sim.py on the bench workload).  Everything rides in
// launches that exist anyway - a forked stream was measured first and costs more than it hides (two cross-stream edges per
// layer in the captured graph: 216 -> 178 tokens/s even when a single chunk is pulled; and the normalise launch runs 5 -> 14
// us while host reads are in flight beside it; profiles/r03_early_fetch.txt):
//   * scan launch: flags the landmark slots whose logit reaches dthr[b][g] (per query head: max_g + ln(k-th value / inv_g)
//     of the PREVIOUS step, written by that step's top-k launch), <= SKV_EARLY_K per 256-slot tile;
//   * normalise launch, ONE extra workgroup per (batch, KV head) - skv_early_prep_role: compacts the flagged slots, looks
//     their chunk ids up, drops the resident ones (LDS bitmap of the resident ids), takes the first E, publishes them
//     (early_ids, early_of[chunk] = staging index).  No host access yet: the normalise launch is latency-bound and slows
//     down 3x beside PCIe reads;
//   * top-k launch, four extra workgroups per (batch, KV head) - skv_early_pull_role: pull the published chunks from the
//     pinned host table into an HBM staging buffer while the top-k (LDS-bound, one CU per head) runs;
//   * fetch launch (skv_rebuild.hip): a miss chunk with early_of[chunk] >= 0 is read from staging instead of the host.
// Nothing here can change a result: staged bytes are the host table's bytes, a wrong prediction costs PCIe bytes only
// (bounded by E), a missing one is fetched as before.
#pragma once
#include "skv_common.h"
#include "skv_select_front.h"
#include "skv_launch.h"
#ifndef PULL_STAMP
#define PULL_STAMP(i)
#endif
#ifdef SKV_TOPK_STAMPS     // the list role's phases, first pull workgroup of the launch (tools/topk_stamps.py)
#define PREP_STAMP(i)                                                                                   \
    do {                                                                                                \
        if (b == 0 && part == 0 && s_list != nullptr && tid == 0) g_topk_stamps[i] = wall_clock64();    \
    } while (0)
#else
#define PREP_STAMP(i)
#endif

#define EF_MAX_E 128
#define EF_MAX_CAND 4096                 // flagged slots examined per head and step (more are dropped)

// inclusive block scan of one int per thread; s_w: THREADS / 64 ints.  Two barriers.
template <int THREADS>
__device__ __forceinline__ int ef_block_scan_incl(int v, int* s_w, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    int add = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w)
        if (w < wave) add += s_w[w];
    __syncthreads();
    return x + add;
}

static inline size_t skv_early_prep_lds_bytes(int n_chunks) {
    return ((size_t)(n_chunks + 31) / 32 + 64 + SKV_EARLY_GAPS) * sizeof(int);   // bitmap | scan scratch | gap table
}

// One workgroup of THREADS threads for (batch, head) b.  R <= 4 * THREADS resident slots, T * SKV_EARLY_K <= EF_MAX_CAND.
// Two dependent round trips (flag entries + resident ids + last step's list, then the slot -> chunk id gathers), one block
// scan.  Candidate i = (tile i / K, entry i % K) is real when its entry index is below the tile's count; a thread's
// candidates are i = c * THREADS + tid, and every load is issued unconditionally (a load under `if` would be followed by
// a wait for its round trip - 16 of them in a row).
// publish = false (fused selection: several workgroups of a head build the SAME list, each for itself - only one of them may
// maintain early_of / early_ids / early_cnt): nothing is written to global memory.  s_list (nullable): the kept ids [E] and,
// in s_list[EF_MAX_E], their number - for a role that goes on in the same workgroup (skv_early_prep_pull_role).
template <int THREADS>
__device__ __forceinline__ void skv_early_prep_role(const EarlyHooks& eh, int b, int tid, int* smem, int part = 0, int parts = 1,
                                                    int* s_list = nullptr) {
    constexpr int CPT = EF_MAX_CAND / THREADS;
    const int n_chunks = eh.n_chunks, E = eh.E, T = eh.T, N = eh.N, R = eh.R;
    const int words = (n_chunks + 31) / 32;
    int* const s_bits = smem;                              // [words] resident chunk ids
    int* const s_w = s_bits + words;                       // [<= 16] scan scratch
    const int total = min(T * SKV_EARLY_K, EF_MAX_CAND);
    // ---- round trip 1.  The workgroup PUBLISHES the staging slots it owns (e % parts == part: the slots its pull role fills -
    // one workgroup owns them all when the list is built in the normalise launch): last step's early_of entries of these
    // slots go back to -1 first (the stores are acknowledged - s_waitcnt below, behind the gathers every thread waits for
    // anyway - before any thread writes a new entry: the scan's barriers lie between).  early_ids[e] = -1: slot e unused.
    const bool own_t = tid < E && tid % parts == part;
    const int prev_id = own_t ? eh.early_ids[(size_t)b * E + tid] : -1;
    const int rounds = (total + THREADS - 1) / THREADS;    // (uniform: 4 at the headline shape, 7 at GLM-4 200K)
    int slot[CPT], tcnt[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        slot[c] = -1;
        tcnt[c] = 0;
        if (c < rounds) {
            const int i = min(c * THREADS + tid, total - 1);
            slot[c] = eh.flag_slot[(size_t)b * T * SKV_EARLY_K + i];
            tcnt[c] = eh.flag_cnt[(size_t)b * T + i / SKV_EARLY_K];
        }
    }
    // slot -> chunk id without the gather (round 4): the head's gap table, in LDS behind the scan scratch
    const bool mapped = eh.gap_slots != nullptr && eh.map_ok[b] != 0;      // (uniform)
    int* const s_gap = s_w + 32;                           // [SKV_EARLY_GAPS]
    if (mapped && tid < SKV_EARLY_GAPS) s_gap[tid] = eh.gap_slots[(size_t)b * SKV_EARLY_GAPS + tid];
    long long my_res[4];                                   // resident ids of slots tid, tid + THREADS, ... (R <= 4 * THREADS)
#pragma unroll
    for (int k = 0; k < 4; ++k) my_res[k] = tid + k * THREADS < R ? eh.resident[(size_t)b * R + tid + k * THREADS] : -1ll;
    // chunks staged AHEAD of this step (near misses of the previous one, skv_near_pull_role) count as resident here: their
    // bytes are in staging already and early_of names them - flagging them again would pull them twice
    const int my_near = (eh.near_pub != nullptr && tid < SKV_NEAR_SLOTS)
                            ? eh.near_pub[((size_t)(tid >> 6) * eh.near_B + b) * SKV_NEAR_MAX + (tid & 63)] : -1;     // (both lists)
    for (int i = tid; i < words; i += THREADS) s_bits[i] = 0;
    // (only an entry that still names this slot: the chunk may have been staged AHEAD since - skv_near_pull_role publishes it under
    // a slot >= E - and that entry must survive; the load is one more request of this round trip, the store follows it)
    if (prev_id >= 0 && prev_id < n_chunks && eh.early_of[(size_t)b * n_chunks + prev_id] == (short)tid)
        eh.early_of[(size_t)b * n_chunks + prev_id] = (short)-1;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (my_res[k] >= 0 && my_res[k] < n_chunks) atomicOr(&s_bits[my_res[k] >> 5], 1 << (my_res[k] & 31));
    if (my_near >= 0 && my_near < n_chunks) atomicOr(&s_bits[my_near >> 5], 1 << (my_near & 31));
    PREP_STAMP(19);
    // ---- round trip 2: slot -> chunk id
    long long id[CPT];
    unsigned realm = 0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int i = c * THREADS + tid;
        const bool real = i < total && (i % SKV_EARLY_K) < min(max(tcnt[c], 0), SKV_EARLY_K) && slot[c] >= 0 && slot[c] < N;
        if (real) realm |= 1u << c;
        id[c] = -1;
        if (c < rounds) {
            if (mapped) {      // id = slot + #{gaps <= slot}: binary search over the ascending table (7 LDS reads)
                const int sl = real ? slot[c] : 0;
                int lo = 0;
#pragma unroll
                for (int w = SKV_EARLY_GAPS / 2; w >= 1; w >>= 1) lo += (s_gap[lo + w - 1] <= sl) ? w : 0;
                id[c] = sl + lo;
            } else {
                id[c] = eh.lm_idx[(size_t)b * N + (real ? slot[c] : 0)];
            }
        }
    }
    PREP_STAMP(20);
    __syncthreads();                                       // the bitmap is complete
    PREP_STAMP(21);
    int nkeep = 0;
    unsigned keepm = 0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        if (((realm >> c) & 1u) && id[c] >= 0 && id[c] < n_chunks && !((s_bits[id[c] >> 5] >> (id[c] & 31)) & 1)) {
            keepm |= 1u << c;
            ++nkeep;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the clears above are in L2)
    PREP_STAMP(22);
    // (1,024 threads: the selection's DPP scan with ONE barrier - the shuffle scan above took 1.1 of the list's 4.2 us, in-step stamps)
    int kincl;
    if constexpr (THREADS == T2_THREADS) kincl = block_scan_incl1(nkeep, s_w, tid);
    else kincl = ef_block_scan_incl<THREADS>(nkeep, s_w, tid);
    PREP_STAMP(23);
    if (tid == THREADS - 1) {
        if (part == 0) eh.early_cnt[b] = min(kincl, E);
        if (s_list) s_list[EF_MAX_E] = min(kincl, E);
    }
    int pos = kincl - nkeep;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        if ((keepm >> c) & 1u) {
            if (pos < E) {
                if (pos % parts == part) {
                    eh.early_ids[(size_t)b * E + pos] = (int)id[c];
                    eh.early_of[(size_t)b * n_chunks + id[c]] = (short)pos;
                }
                if (s_list) s_list[pos] = (int)id[c];
            }
            ++pos;
        }
    }
    if (own_t) {                                           // owned slots behind the end of the list are empty
        int n_all = 0;                                     // (the scan's wave totals are still in s_w: its last barrier is behind us)
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) n_all += s_w[w];
        if (tid >= min(n_all, E)) eh.early_ids[(size_t)b * E + tid] = -1;
    }
}

// Pull role: eh.pull_wgs workgroups of THREADS threads per (batch, head) b (SKV_EARLY_PULL_WGS for one sequence, 1 for batches,
// which have a head per CU and more anyway), workgroup `part` takes the published chunks e = part, part + pull_wgs, ...  (How fast host memory can be read depends on how many CUs ask: one workgroup per
// head - 4 CUs at GLM-4's shape - pulled 393 KB in 15 us = 26 GB/s and stretched the top-k launch from 12.7 to 18.5 us; with
// four per head it ends with the top-k, 12.6 us.)  s_sel: EF_MAX_E ints.
#define SKV_EARLY_PULL_WGS 4
// list_in_lds: s_sel[0 .. E) and s_sel[EF_MAX_E] were filled by skv_early_prep_role in this workgroup (fused selection)
template <int THREADS>
__device__ __forceinline__ void skv_early_pull_role(const EarlyHooks& eh, int b, int part, int tid, int* s_sel, bool list_in_lds = false) {
    const int E = eh.E;
    if (!list_in_lds && tid < EF_MAX_E) s_sel[tid] = tid < E ? eh.early_ids[(size_t)b * E + tid] : 0;
    __syncthreads();
    const int n_sel = list_in_lds ? min(s_sel[EF_MAX_E], E) : min(eh.early_cnt[b], E);
    // chunk e = 128 units of 16 B; 8 requests per thread in flight (unconditional loads through a selected pointer: a load
    // under `if` would be followed by a wait for the PCIe round trip)
    const u32x4* const hb = reinterpret_cast<const u32x4*>(eh.v_host) + (long long)b * eh.v_host_stride_u128;
    u32x4* const sb = reinterpret_cast<u32x4*>(eh.staging) + (size_t)b * eh.stage_stride * 128;   // (slots [E, stride): near misses)
    const int PW = eh.pull_wgs;
    const int mine = (n_sel - part + PW - 1) / PW;           // chunks of this workgroup
    for (int r0 = 0; r0 * THREADS < mine * 128; r0 += 8) {
        u32x4 v[8];
        int dst[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = (r0 + k) * THREADS + tid, e = (idx >> 7) * PW + part, u = idx & 127;
            const bool on = e < n_sel;
            dst[k] = on ? e * 128 + u : -1;
            const u32x4* src = on ? hb + (long long)s_sel[e] * 128 + u : reinterpret_cast<const u32x4*>(sb) + u;
            v[k] = *src;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (dst[k] >= 0) sb[dst[k]] = v[k];
    }
}

// Fused selection: there is no normalise launch for the list role to ride in, so every pull workgroup of a head builds the
// head's list itself (~3 us: two dependent round trips and a block scan), pulls its share - the staging slots e with
// e % pull_wgs == part - and publishes EXACTLY those slots for the fetch launch (early_ids[e], early_of[chunk] = e; workgroup 0
// also the count, a diagnostic).  The list's "resident" input is the slot -> chunk map the selection workgroup of the same
// launch rewrites at ITS end (~10 us later): should a pull workgroup ever start so late that it sees some of the new ids, its
// list differs from the other workgroups' - which is why no workgroup publishes another one's slots: early_of[c] = e is only
// ever written by the workgroup that also writes staging[e], from the same list, so a staged chunk is the chunk its entry
// names whatever the lists were; differing lists cost duplicate or missing pulls (PCIe bytes), never a result.  (A reset
// by one workgroup racing a new entry of the same chunk by another - a chunk predicted in two consecutive steps that changes
// its slot: rare, a predicted chunk is nearly always selected and then resident - leaves -1 or the new slot, in either order
// (the workgroups sit on different XCDs, their L2s write back at the end of the launch): both are valid, -1 re-reads the
// chunk from the host.)
// smem: [words(n_chunks) + 64 + SKV_EARLY_GAPS] ints of the list role, then EF_MAX_E + 1 ints of the list.
template <int THREADS>
__device__ __forceinline__ void skv_early_prep_pull_role(const EarlyHooks& eh, int b, int part, int tid, int* smem) {
#ifdef SKV_TOPK_STAMPS
    const bool pull_stamp_wg = b == 0 && part == 0;
#endif
    PULL_STAMP(24);
    int* const s_list = smem + (eh.n_chunks + 31) / 32 + 64 + SKV_EARLY_GAPS;
    if (tid <= EF_MAX_E) s_list[tid] = 0;
    __syncthreads();
    skv_early_prep_role<THREADS>(eh, b, tid, smem, part, eh.pull_wgs, s_list);
    PULL_STAMP(25);
    // (Measured and dropped, profiles/r04_fused_selection.txt: holding the first host read back until the selection workgroups'
    // two dependent device-memory round trips are through - device-memory latency stretches chip-wide while host reads are in
    // flight - by a fixed 3 / 5 / 7 / 9 us from the workgroup's start: 220.4 / 220.5 / 218.0 / 215.8 tokens/s against 220.9 with
    // no delay at the headline shape, 189.1 against 194.2 at GLM-4's: the link time lost outweighs the stretch.)
    __syncthreads();
    skv_early_pull_role<THREADS>(eh, b, part, tid, s_list, true);
#ifdef SKV_TOPK_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PULL_STAMP(26);
}

// ---------------------------------------------------------------------------------------------------------------------
// Near-miss staging for the NEXT step (round 5).  The link idles during the 79 us of dense GEMVs of a
// layer; which chunks the next step will miss is not known yet, but a third of the chunks that fell just short of THIS
// step's selection are selected next (tools/near_miss_sim.py, bench workload: 64 nearest -> 19.5 of the ~83 misses per head).
// `parts` (1, 2 or 4) workgroups of 256 threads per (batch, head), riding as the FIRST workgroups of the layer's gate/up GEMV
// launch (no extra launch or stream - the form that lost 13 % in round 3).  Part p owns the chunks with id % parts == p and the
// staging slots E + p * 64 / parts .. : it reconciles its share of the new near-miss list (near_ids, from this
// step's top-k launch) with what its slots hold (near_pub), keeps what is in both, and pulls the rest from the pinned host
// table into the slots of chunks that dropped out of the list - independent of the other parts (no workgroup reads state
// another one writes in the same launch).  One workgroup per head pulling slowly is the form that pays (skv_launch.h): the
// harder the role pulls, the more the HBM-bound GEMV beside it loses.  early_of[chunk] = E + slot is the map
// the in-step early fetch publishes in too, so the fetch launch of the next step finds these chunks with the lookup it does
// anyway, and the in-step list leaves them alone (they are in its resident bitmap).
// Nothing here can change a result: a staging slot is published (near_pub, early_of) only with the host table's bytes of
// the chunk it names; a chunk's old entry is cleared before its slot is overwritten, both by this workgroup, in this order,
// behind a barrier; everything is consumed by LATER launches.  An entry is only cleared if it still names this slot.
// smem: 6 * 64 + 8 ints.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef SKV_NEAR_PULL_CAP
#define SKV_NEAR_PULL_CAP 32       // chunks one workgroup of the pull role stages per launch = ONE round of requests (the rest of the list waits a step; measured: 64 / 32 / 16 -> 230.9 / 231.7 / 228.3 tokens/s, 226.8 without the role)
#endif
#ifndef SKV_NEAR_INFLIGHT
#define SKV_NEAR_INFLIGHT 16       // 16-B requests a thread of the pull role keeps in flight (32: 188 VGPRs - a GEMV wave per SIMD less)
#endif
__device__ __forceinline__ void skv_near_pull_role(const NearPull& np, int blk, int tid, int* smem) {
    constexpr int NL = SKV_NEAR_MAX;
    static_assert(NL == 64, "one wave reconciles: the list and a part's slots in 64-bit ballots");
    const int PARTS = np.parts, PS = NL / PARTS;                           // (parts in {1, 2, 4}: checked by the launcher)
    const int b = blk / PARTS, part = blk % PARTS;
    int* const s_new = smem;              // [NL] this step's near misses of this part (-1: none / another part's)
    int* const s_old = smem + NL;         // [<= 64] staged now in this part's slots
    int* const s_slot = s_old + 64;       // [<= 64] local slot of the i-th chunk to pull
    int* const s_id = s_slot + 64;        // [<= 64] its id
    int* const s_n = s_id + 64;           // [8] chunks to pull
    int* const s_onew = s_n + 8;          // [NL] the other list's near misses of this step (kept if staged here)
    int* const s_opub = s_onew + NL;      // [NL] what the other list's slots hold (not pulled again)
    const int n_chunks = np.n_chunks, E = np.E;
    const int slot0 = part * PS;          // first near slot of this part
    if (tid < NL) {
        const int cnt = min(max(np.near_cnt[b], 0), NL);
        int n = tid < cnt ? np.near_ids[(size_t)b * NL + tid] : -1;
        if (n < 0 || n >= n_chunks || n % PARTS != part) n = -1;
        s_new[tid] = n;
        if (tid < PS) {
            const int o = np.near_pub[(size_t)b * NL + slot0 + tid];
            s_old[tid] = (o >= 0 && o < n_chunks) ? o : -1;
        }
        const int ocnt = np.other_cnt != nullptr ? min(max(np.other_cnt[b], 0), NL) : 0;
        s_onew[tid] = tid < ocnt ? np.other_ids[(size_t)b * NL + tid] : -2;      // (-2: matches no id and no empty slot)
        s_opub[tid] = np.other_pub != nullptr ? np.other_pub[(size_t)b * NL + tid] : -2;
    }
    __syncthreads();
    if (tid < NL) {                       // wave 0
        const int n = s_new[tid], o = tid < PS ? s_old[tid] : -1;
        bool keep = false, fresh = n >= 0;
#pragma unroll 8
        for (int i = 0; i < NL; ++i) {
            keep |= o >= 0 && (s_new[i] == o || s_onew[i] == o);    // (wanted by either list: stays where it is)
            if (i < tid) fresh &= s_new[i] != n;          // (the list holds distinct ids; a duplicate would be staged once)
            fresh &= s_opub[i] != n;                       // (staged by the other list's launch already)
        }
        for (int j = 0; j < PS; ++j) fresh &= s_old[j] != n;
        const unsigned long long slots_m = PS >= 64 ? ~0ull : ((1ull << (PS & 63)) - 1ull);
        const unsigned long long free_m = ~__ballot(tid < PS && keep) & slots_m;   // slots whose chunk left the list (or empty)
        const unsigned long long fresh_m = __ballot(fresh);
        const int n_take = min(min(__builtin_popcountll(fresh_m), __builtin_popcountll(free_m)), SKV_NEAR_PULL_CAP);   // (the list's tail waits)
        if (fresh) {
            const int k = __builtin_popcountll(fresh_m & ((1ull << tid) - 1ull));
            if (k < n_take) {
                unsigned long long m = free_m;                             // the k-th free slot
                for (int i = 0; i < k; ++i) m &= m - 1;
                const int e = __builtin_ctzll(m);
                s_slot[k] = e;
                s_id[k] = n;
                const int old = s_old[e];
                // the slot's old chunk: unpublished before its bytes are overwritten (only if the entry still names this slot)
                if (old >= 0 && np.early_of[(size_t)b * n_chunks + old] == (short)(E + np.slot_base + slot0 + e))
                    np.early_of[(size_t)b * n_chunks + old] = (short)-1;
            }
        }
        if (tid == 0) s_n[0] = n_take;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int n_pull = s_n[0];
    const u32x4* const hb = reinterpret_cast<const u32x4*>(np.v_host) + (long long)b * np.v_host_stride_u128;
    u32x4* const sb = reinterpret_cast<u32x4*>(np.staging) + ((size_t)b * (E + SKV_NEAR_SLOTS) + E + np.slot_base + slot0) * 128;
    // chunk = 128 units of 16 B; 256 threads: two chunks per pass, SKV_NEAR_INFLIGHT requests per thread in flight
    // (unconditional loads through a selected pointer, as in skv_early_pull_role)
    for (int r0 = 0; r0 * 256 < n_pull * 128; r0 += SKV_NEAR_INFLIGHT) {
        u32x4 v[SKV_NEAR_INFLIGHT];
#pragma unroll
        for (int k = 0; k < SKV_NEAR_INFLIGHT; ++k) {
            const int idx = (r0 + k) * 256 + tid, i = idx >> 7, u = idx & 127;
            const u32x4* src = i < n_pull ? hb + (long long)s_id[i] * 128 + u : reinterpret_cast<const u32x4*>(sb) + u;
            v[k] = *src;
        }
#pragma unroll
        for (int k = 0; k < SKV_NEAR_INFLIGHT; ++k) {
            const int idx = (r0 + k) * 256 + tid, i = idx >> 7, u = idx & 127;
            if (i < n_pull) sb[s_slot[i] * 128 + u] = v[k];
        }
    }
    if (tid < n_pull) {                   // publish (read by later launches only)
        np.near_pub[(size_t)b * NL + slot0 + s_slot[tid]] = s_id[tid];
        np.early_of[(size_t)b * n_chunks + s_id[tid]] = (short)(E + np.slot_base + slot0 + s_slot[tid]);
    }
}
