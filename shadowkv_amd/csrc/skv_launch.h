// Host-side launch descriptors shared between the API layer (skv_api.hip) and the kernel files.
#pragma once
#include <stdint.h>
#include <stddef.h>

// attention role of the fused in-place fetch launch (skv_rebuild.hip)
struct AttnLaunch {
    const void* q;           // [bs][Hq][128] bf16
    void* ws;                // attention workspace, rec_splits records per query head
    const int* kv_len_dev;   // nullable
    int kv_len_host;
    int kv_rows;             // rows a head owns in the cache buffers: the device-side kv_len is clamped to it
    int G, splits, rec_splits;
    float scale;
    int resident_sets;       // slots of the sparse region (>= select_sets; the generated rows sit behind resident_sets * 8 rows)
    // speculative early V fetch (skv_early.hip), all null / 0 = off: a miss chunk c with early_of[bh][c] = e >= 0 has its
    // 2 KiB in early_staging[bh][e] already - the landing waves read it from there instead of the host table
    const short* early_of;   // [bs*heads][early_chunks]
    const void* early_staging;
    int early_chunks, early_max;
};

// Consumer side of the speculative early V fetch for the launches without the attention role (skv_fetch_kv*): same meaning as
// the early_* fields of AttnLaunch.
struct EarlyConsume {
    const short* early_of;
    const void* early_staging;
    int early_chunks, early_max;
};

// Fused selection (round 4; skv_select.hip "logit-domain prefilter"): the scan launch leaves, per landmark slot, a 15-bit
// order-preserving KEY of kappa_j = max_g (logit_gj - ctil_g) - the group maximum in the logit domain under the PREVIOUS step's
// log-normalisers ctil_g - and the slot's G logits contiguously (Dt [B][N][G]) instead of D [B][G][N]; the top-k launch then
// finds the exact top-S without a normalise launch (see skv_topk3_kernel).  ctil null = the three-launch path.
struct FusedSel {
    const float* ctil;        // [B][G] in: log-normalisers (m_g + ln s_g) the keys are taken against (any finite values are valid)
    uint16_t* keys;           // [B][key_stride] out
    void* Dt;                 // [B][N][G] bf16 out
    int key_stride;
};

// the top-k launch's side of it
struct FusedTop {
    const void* Dt;           // [B][N][G] bf16 (scan launch, FusedSel); null = scores in (three-launch path)
    const float* part_max;    // [B][T][G]
    const float* part_sum;
    float* ctil;              // [B][G] in: what the scan's keys were taken against; out: this step's c_g
    int* level;               // [B] in / out: witness level - a key that at least S keys of the row are expected to reach (0: none yet)
    int* stats;               // [B][2] out: path of this launch (bit 0: the level was searched, bit 1: every slot evaluated), candidates
    int T;
};

// Hooks of the speculative early V fetch in the selection launches (skv_select.hip, roles in skv_early.h); dthr_in null = off.
//   scan:      a landmark slot whose logit reaches dthr_in[b][g] for some query head g is FLAGGED (it would have made the
//              previous step's top-k): per tile the first SKV_EARLY_K flagged slots go to flag_slot, their number to flag_cnt.
//              Prediction only - the selection itself never reads it.
//   normalise: workgroup 0 of a head leaves the rows' finals (max, 1 / sum) in finals[b][g][2]; one extra workgroup per head
//              turns the flags into the list of chunks to pull (skv_early_prep_role).
//   top-k:     dthr_out[b][g] for the NEXT step = max_g + ln(k-th value / inv_g) + margin  (P >= k-th value <=> D >= this);
//              one extra workgroup per head pulls the listed chunks into the staging buffer (skv_early_pull_role).
struct EarlyHooks {
    const float* dthr_in;
    int* flag_cnt;
    int* flag_slot;
    float* finals;
    float* dthr_out;
    int G;
    float margin;
    const int64_t* lm_idx;      // [B][N] slot -> chunk id
    const int64_t* resident;    // [B][R] resident chunk ids
    int* early_cnt;
    int* early_ids;
    short* early_of;
    const void* v_host;
    long long v_host_stride_u128;
    void* staging;
    int T, N, R, n_chunks, E;
    int pull_wgs;               // pull workgroups per (batch, head) in the top-k launch (4 for one sequence, 1 for batches)
    // slot -> chunk id WITHOUT the gather (skv_early_state_set_landmark_map): the landmark ids of a head are the chunk ids in
    // ascending order minus a few (the outlier chunks), so id(slot) = slot + #{i : gap_slots[i] <= slot}; map_ok[b] == 0: gather
    const int* gap_slots;       // [B][SKV_EARLY_GAPS] ascending, INT_MAX padded
    const int* map_ok;          // [B]
    // near-miss staging for the NEXT step (round 5, see EarlyState): the top-k launch leaves up to SKV_NEAR_MAX chunk ids that
    // fell just short of this step's selection (near_ids / near_cnt, prediction only); the pull role of the gate/up GEMV
    // launch stages them (skv_near_pull_role) and publishes them in early_of with staging indices E .. E + SKV_NEAR_SLOTS - 1;
    // near_pub = the ids published there now (-1: slot unused) - the in-step list treats them like resident chunks.
    // SKV_NEAR_LISTS lists: list k holds the candidates ranked S + 64 k + 1 .. S + 64 (k + 1) and owns staging slots
    // E + 64 k .. E + 64 k + 63; list 0 is staged by the gate/up launch, list 1 by the down-projection launch.
    int* near_cnt;              // [LISTS][B]
    int* near_ids;              // [LISTS][B][SKV_NEAR_MAX]
    const int* near_pub;        // [LISTS][B][SKV_NEAR_MAX]
    int stage_stride;           // staging slots per (batch, head): E + SKV_NEAR_SLOTS
    int near_B;                 // B (stride of the list index)
};

// Speculative early V fetch (round 3; skv_early.hip).  One state buffer per layer (skv_early_state_bytes), carved here.
//   dthr      [B][G]    f32   logit threshold of query head g for the NEXT step's scan: a landmark slot is flagged when some
//                              head's logit reaches it (= the slot would have made the previous step's top-k); +inf: none
//   finals    [B][G][2] f32   (max, 1 / sum) of this step's softmax rows, left by the normalise launch for the top-k launch
//   flag_cnt  [B][T]    i32   flagged slots of tile t (<= SKV_EARLY_K kept), flag_slot [B][T][SKV_EARLY_K] the slots
//   early_cnt [B]       i32   chunks the early launch pulled this step, early_ids [B][E] their chunk ids
//   early_of  [B][chunks] i16 staging index of a chunk pulled early this step, -1 otherwise
//   staging   [B][E][2 KiB]   the pulled V chunks
//   near_cnt  [B] i32, near_ids [B][SKV_NEAR_MAX] i32   round 5: chunks just below this step's selection (written by the top-k
//                              launch), candidates for staging ahead of the NEXT step; near_pub [B][SKV_NEAR_SLOTS] i32 the chunks
//                              staged that way now: chunk near_pub[b][e] sits in staging slot E + e, early_of[chunk] = E + e
//   staging is [B][E + SKV_NEAR_SLOTS][2 KiB]: slots [0, E) belong to the in-step early fetch, [E, E + SKV_NEAR_SLOTS) to the near misses
#define SKV_EARLY_K 16
#define SKV_EARLY_GAPS 128     // chunks that may be missing from a head's ascending landmark-id sequence (outliers: 24 per 1024 budget)
#define SKV_NEAR_MAX 64        // near misses the top-k launch lists per (batch, head)
// near-miss staging slots per (batch, head).  The gate/up launch's pull role runs `parts` (1, 2 or 4, a launch argument)
// workgroups per (batch, head): part p owns the chunks with id % parts == p and the slots [p * SLOTS / parts, (p + 1) * SLOTS / parts).
// Measured on MI355X, tokens/s with / without the role on one box (profiles/r05_near_fetch.txt):
//   8 KV heads (Llama-3.1-8B 122K): 1 part, 16 requests per thread in flight: 233.9 / 226.9 (+3.0 %); 4 parts: 227.3 / 226.7 (+0.3 %)
// - the role costs the HBM-bound GEMV it rides in more the harder it pulls: a slow trickle from ~8 CUs is what hides.
#define SKV_NEAR_LISTS 2
#define SKV_NEAR_SLOTS (SKV_NEAR_LISTS * 64)
struct EarlyState {
    float* dthr;
    float* finals;
    int* flag_cnt;
    int* flag_slot;
    int* early_cnt;
    int* early_ids;
    short* early_of;
    void* staging;
    int* gap_slots;
    int* map_ok;
    int* near_cnt;
    int* near_ids;
    int* near_pub;
    size_t total;
};
static inline size_t skv_early_align(size_t x) { return (x + 255) & ~(size_t)255; }
static inline EarlyState skv_carve_early(void* base, int B, int G, int n_landmarks, int n_chunks, int E) {
    const size_t T = (size_t)(n_landmarks + 255) / 256;
    unsigned char* p = (unsigned char*)base;
    size_t off = 0;
    EarlyState e;
    e.dthr = (float*)(p + off);      off += skv_early_align((size_t)B * G * 4);
    e.finals = (float*)(p + off);    off += skv_early_align((size_t)B * G * 8);
    e.flag_cnt = (int*)(p + off);    off += skv_early_align((size_t)B * T * 4);
    e.flag_slot = (int*)(p + off);   off += skv_early_align((size_t)B * T * SKV_EARLY_K * 4);
    e.early_cnt = (int*)(p + off);   off += skv_early_align((size_t)B * 4);
    e.early_ids = (int*)(p + off);   off += skv_early_align((size_t)B * E * 4);
    e.early_of = (short*)(p + off);  off += skv_early_align((size_t)B * n_chunks * 2);
    e.staging = p + off;             off += skv_early_align((size_t)B * (E + SKV_NEAR_SLOTS) * 2048);
    e.gap_slots = (int*)(p + off);   off += skv_early_align((size_t)B * SKV_EARLY_GAPS * 4);
    e.map_ok = (int*)(p + off);      off += skv_early_align((size_t)B * 4);
    e.near_cnt = (int*)(p + off);    off += skv_early_align((size_t)SKV_NEAR_LISTS * B * 4);
    e.near_ids = (int*)(p + off);    off += skv_early_align((size_t)SKV_NEAR_LISTS * B * SKV_NEAR_MAX * 4);
    e.near_pub = (int*)(p + off);    off += skv_early_align((size_t)SKV_NEAR_LISTS * B * SKV_NEAR_MAX * 4);
    e.total = off;
    return e;
}
// the selection launches' view of a carved state (one place: every entry that takes an early state builds its hooks here)
static inline EarlyHooks skv_early_hooks(const EarlyState& es, int blocks, int groups, float margin, const int64_t* landmark_idx,
                                         const int64_t* resident, const void* v_host, long long host_block_stride,
                                         int n_landmarks, int resident_sets, int n_chunks, int E) {
    EarlyHooks eh{es.dthr, es.flag_cnt, es.flag_slot, es.finals, es.dthr, groups, margin, landmark_idx, resident,
                  es.early_cnt, es.early_ids, es.early_of, v_host, host_block_stride / 8, es.staging,
                  (n_landmarks + 255) / 256, n_landmarks, resident_sets, n_chunks, E};
    eh.gap_slots = es.gap_slots;
    eh.map_ok = es.map_ok;
    eh.near_cnt = es.near_cnt;
    eh.near_ids = es.near_ids;
    eh.near_pub = es.near_pub;
    eh.stage_stride = E + SKV_NEAR_SLOTS;
    eh.near_B = blocks;
    return eh;
}

// Near-miss staging ahead of the next step (round 5): the pull role that rides in the gate/up GEMV launch of a layer
// (skv_gemv.hip: the first `blocks` workgroups of that launch; skv_near_pull_role in skv_early.h).
struct NearPull {
    const int* near_cnt;        // [B]   (top-k launch of this step; the launch's list)
    const int* near_ids;        // [B][SKV_NEAR_MAX]
    int* near_pub;              // [B][SKV_NEAR_MAX] in / out: what the list's staging slots hold
    short* early_of;            // [B][n_chunks]
    void* staging;              // [B][E + SKV_NEAR_SLOTS][2 KiB]
    const void* v_host;
    long long v_host_stride_u128;
    int blocks;                 // B * parts pull workgroups (B = batch x KV heads)
    int n_chunks, E, parts;
    int slot_base;              // first staging slot of the list behind E: 64 * list
    // the OTHER list (staged by another launch of the same step): a chunk it holds is not pulled again, a chunk it wants is kept
    const int* other_cnt;       // [B]   (all three null: only this list is staged)
    const int* other_ids;       // [B][SKV_NEAR_MAX]
    const int* other_pub;       // [B][SKV_NEAR_MAX]
};
static inline NearPull skv_near_pull(const EarlyState& es, const void* v_host, long long host_block_stride, int B, int n_chunks, int E,
                                     int parts, int list, int active_lists) {
    if (active_lists < 2)       // the other list is not staged by anybody: nothing of it to keep or to skip
        return NearPull{es.near_cnt + (size_t)list * B, es.near_ids + (size_t)list * B * SKV_NEAR_MAX, es.near_pub + (size_t)list * B * SKV_NEAR_MAX,
                        es.early_of, es.staging, v_host, host_block_stride / 8, B * parts, n_chunks, E, parts, 64 * list, nullptr, nullptr, nullptr};
    return NearPull{es.near_cnt + (size_t)list * B, es.near_ids + (size_t)list * B * SKV_NEAR_MAX, es.near_pub + (size_t)list * B * SKV_NEAR_MAX,
                    es.early_of, es.staging, v_host, host_block_stride / 8, B * parts, n_chunks, E, parts, 64 * list,
                    es.near_cnt + (size_t)(1 - list) * B, es.near_ids + (size_t)(1 - list) * B * SKV_NEAR_MAX,
                    es.near_pub + (size_t)(1 - list) * B * SKV_NEAR_MAX};
}
