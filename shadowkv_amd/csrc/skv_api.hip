// extern "C" surface of libshadowkv_hip.so (declared in include/shadowkv_hip.h).
// Thin argument checking + dispatch to the launchers in skv_select / skv_move / skv_rebuild /
// skv_attn / skv_rope.  No allocation, no synchronisation: every entry point is graph-capturable.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"
#include "skv_launch.h"

#include <string.h>

// launchers (defined in the kernel files)
int skv_launch_score(const void* q, const void* lm, void* D, float* pmax, float* psum, int B, int G, int N,
                     float alpha, hipStream_t st, const EarlyHooks* hooks = nullptr, const FusedSel* fused = nullptr);
int skv_launch_softmax_final_apply(const void* D, float* pmax, float* psum, void* P, int B, int m, int N,
                                   hipStream_t st);
int skv_launch_normalize_groupmax(const void* D, const float* pmax, const float* psum, void* P, void* score,
                                  int score_stride, int B, int G, int N, hipStream_t st, const EarlyHooks* hooks = nullptr);
int skv_launch_topk_reorder(const void* score, int score_stride, const int64_t* lm_idx, const int64_t* cur_in,
                            int64_t* cached, int32_t* offsets, int32_t* cnts, int64_t* sel_out, int32_t* dst_slots,
                            int B, int N, int S, hipStream_t st);
int skv_launch_topk_resident(const void* score, int score_stride, const int64_t* lm_idx, const int64_t* cur_in,
                             int64_t* cached, int32_t* offsets, int32_t* cnts, int64_t* sel_out, int32_t* dst_slots,
                             int B, int N, int S, int R, int32_t* slot_age, hipStream_t st, const EarlyHooks* hooks = nullptr,
                             const FusedTop* fused = nullptr, int G = 0);
bool skv_fused_select_supported(int G, int N, int S);
int skv_launch_early_init(const EarlyState& es, int B, int G, int n_landmarks, int n_chunks, int E, hipStream_t st);
int skv_launch_early_map(const EarlyState& es, const int64_t* lm_idx, int B, int N, hipStream_t st);
int skv_launch_move_rows(const void* host_rows, void* dev, void* temp, const int32_t* offsets, const int32_t* cnts,
                         long long host_len_elems, long long dev_stride_elems, long long dev_off_elems, int B, int S,
                         hipStream_t st);
int skv_launch_gather_rows(const void* host_rows, void* dev, const int64_t* ids, long long host_len_elems,
                           long long dev_len_elems, int B, int S, hipStream_t st);
int skv_launch_rebuild(const void* U, const void* SV, const void* cos_sin, const void* ids, int ids64,
                       const int32_t* cnts, void* out, int bs, int heads, int seq_len, int head_dim, int R, int S,
                       int C, long long cs_stride, long long out_stride_b, long long out_stride_h,
                       long long out_stride_s, int out_row0, int mode, const void* hit_temp, const int32_t* hit_offsets,
                       const int32_t* dst_slots, const void* v_host, void* v_buf, const void* v_temp,
                       long long v_host_stride, long long v_stride, long long v_off, hipStream_t st,
                       const AttnLaunch* attn, const EarlyConsume* early = nullptr);
int skv_launch_attn_merge(const void* ws, const int32_t* cnts, void* out, int bs, int Hq, int Hkv, int S, int splits,
                          hipStream_t st);
int skv_launch_stage_hits(void* k_buf, void* k_temp, void* v_buf, void* v_temp, const int32_t* offsets,
                          const int32_t* cnts, long long stride_elems, long long off_elems, int B, int S,
                          hipStream_t st);
int skv_launch_land_rows(const void* host_rows, void* buf, const void* temp, const int32_t* offsets,
                         const int32_t* cnts, long long host_len_elems, long long stride_elems, long long off_elems,
                         int B, int S, hipStream_t st);
int skv_launch_sparse_attention(const void* q, const void* k, const void* v, void* out, void* ws,
                                const int* kv_len_dev, int kv_len_host, int kv_rows, long long kv_stride_h, int bs, int Hq,
                                int Hkv, int head_dim, int splits, float scale, const int32_t* slots, int n_slots,
                                int sparse_start, int resident_rows, hipStream_t st);
int skv_launch_rope_chunked(const void* x, const void* cos_sin, const int32_t* pid, void* out, const int32_t* cnts,
                            int batch, int heads, int seq_len, int embed_dim, long long sxb, long long sxh,
                            long long sxs, long long sxe, long long scs, long long spb, long long sph, long long sps,
                            long long sob, long long soh, long long sos, int off_start, int off_end, int half_dim,
                            int chunk, int glm, int mode, hipStream_t st);
int skv_launch_rope_plain(const void* x, const void* cos_sin, const void* sin, long long ssin, const int64_t* pid,
                          void* out, int batch, int heads, int seq_len, int embed_dim, long long sxb, long long sxh,
                          long long sxs, long long sxe, long long scs, long long spb, long long sph, long long sps,
                          int half_dim, hipStream_t st);

static thread_local char g_err[256] = "";

static int finish(int rc) {
    if (rc != SKV_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        strncpy(g_err, hipGetErrorString(e), sizeof(g_err) - 1);
        return SKV_ERR_LAUNCH;
    }
    return SKV_OK;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int skv_host_alloc(void** out, size_t nbytes) {
    if (!out || nbytes == 0) return SKV_ERR_ARG;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, nbytes, hipHostMallocDefault);
    if (e != hipSuccess || !*out) {
        strncpy(g_err, hipGetErrorString(e), sizeof(g_err) - 1);
        (void)hipGetLastError();
        return SKV_ERR_LAUNCH;
    }
    return SKV_OK;
}

extern "C" int skv_host_free(void* p) {
    if (!p) return SKV_OK;
    return hipHostFree(p) == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

struct SelectWs {
    void* D;
    float* pmax;
    float* psum;
    void* score;
    int score_stride;
    size_t total;
};
static SelectWs carve_select_ws(void* base, int blocks, int groups, int n) {
    const size_t T = (size_t)(n + 255) / 256;
    SelectWs w;
    size_t off = 0;
    unsigned char* p = (unsigned char*)base;
    w.D = p + off;
    off += align256((size_t)blocks * groups * n * 2);
    w.pmax = (float*)(p + off);
    off += align256((size_t)blocks * T * groups * 4);
    w.psum = (float*)(p + off);
    off += align256((size_t)blocks * T * groups * 4);
    w.score = p + off;
    w.score_stride = (n + 7) & ~7;  // rows start 16-B aligned for the vector staging loads
    off += align256((size_t)blocks * w.score_stride * 2);
    w.total = off;
    return w;
}

extern "C" {

int skv_abi_version(void) { return 1; }
const char* skv_last_error(void) { return g_err; }

// ---------------------------------------------------------------- part 1: legacy surface
int skv_batch_gemm_softmax(const void* A, const void* B, void* D, float* Norm, float* Sum, void* Softmax,
                           int batch_count, int m, int n, int k, float alpha, float beta, skv_stream_t stream) {
    (void)beta;
    if (!A || !B || !D || !Norm || !Sum || !Softmax || batch_count < 1 || n < 1) return SKV_ERR_ARG;
    if (k != 128) return SKV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int rc = skv_launch_score(A, B, D, Norm, Sum, batch_count, m, n, alpha, st);
    if (rc != SKV_OK) return rc;
    return finish(skv_launch_softmax_final_apply(D, Norm, Sum, Softmax, batch_count, m, n, st));
}

int skv_reorder_keys_and_compute_offsets(int64_t* cached_pos_ids, const int64_t* cur_pos_ids, int32_t* offsets,
                                         int32_t* cnts, int batch_size, int heads, int map_size,
                                         skv_stream_t stream) {
    if (!cached_pos_ids || !cur_pos_ids || !offsets || !cnts || batch_size * heads < 1) return SKV_ERR_ARG;
    return finish(skv_launch_topk_reorder(nullptr, 0, nullptr, cur_pos_ids, cached_pos_ids, offsets, cnts, nullptr, nullptr,
                                          batch_size * heads, 0, map_size, (hipStream_t)stream));
}

int skv_gather_copy_with_offsets(const void* values, void* v_cache_buffer, void* temp, const int32_t* offsets,
                                 const int32_t* cnts, uint32_t* signals, int batch_size, int heads,
                                 int cpu_v_length, int gpu_v_length, int gpu_v_offset, int gpu_v_stride,
                                 int map_size, skv_stream_t stream) {
    (void)signals;   // nothing spins: the launch boundary between stage and land is the only ordering
    (void)gpu_v_length;
    if (!values || !v_cache_buffer || !temp || !offsets || !cnts) return SKV_ERR_ARG;
    return finish(skv_launch_move_rows(values, v_cache_buffer, temp, offsets, cnts, cpu_v_length, gpu_v_stride,
                                       gpu_v_offset, batch_size * heads, map_size, (hipStream_t)stream));
}

int skv_gather_copy_d2d_with_offsets(void* keys, const int32_t* offsets, const int32_t* cnts, void* temp,
                                     int batch_size, int heads, int gpu_k_length, int gpu_k_offset,
                                     int gpu_k_stride, int map_size, skv_stream_t stream) {
    (void)gpu_k_length;
    if (!keys || !offsets || !cnts || !temp) return SKV_ERR_ARG;
    return finish(skv_launch_move_rows(nullptr, keys, temp, offsets, cnts, 0, gpu_k_stride, gpu_k_offset,
                                       batch_size * heads, map_size, (hipStream_t)stream));
}

int skv_gather_copy(const void* values, void* v_cache_buffer, const int64_t* position_ids, int batch_size,
                    int heads, int cpu_v_length, int gpu_v_length, int map_size, skv_stream_t stream) {
    if (!values || !v_cache_buffer || !position_ids) return SKV_ERR_ARG;
    return finish(skv_launch_gather_rows(values, v_cache_buffer, position_ids, cpu_v_length, gpu_v_length,
                                         batch_size * heads, map_size, (hipStream_t)stream));
}

int skv_batch_gather_gemm(const void* a, const void* b, const void* cos, const void* sin,
                          const int32_t* position_ids, void* output, int batch_size, int heads, int seq_len,
                          int embed_dim, int rank, int sparse_budget, int max_seq_len, int chunk_size,
                          const int32_t* offset_array, skv_stream_t stream) {
    (void)cos;
    (void)sin;
    (void)max_seq_len;
    if (!a || !b || !position_ids || !output || chunk_size < 1 || sparse_budget % chunk_size) return SKV_ERR_ARG;
    return finish(skv_launch_rebuild(a, b, nullptr, position_ids, 0, offset_array, output, batch_size, heads,
                                     seq_len, embed_dim, rank, sparse_budget / chunk_size, chunk_size, 0,
                                     (long long)heads * sparse_budget * embed_dim,
                                     (long long)sparse_budget * embed_dim, embed_dim, 0, 0, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, nullptr, 0, 0, 0, (hipStream_t)stream, nullptr));
}

#define SKV_ROPE_PUSH_ARGS                                                                                       \
    x, cos_sin, position_ids, output_cache, cnts, batch_size, heads, seq_len, embed_dim, stride_xb, stride_xh,   \
        stride_xs, stride_xe, stride_cos_sin, stride_pid_b, stride_pid_h, stride_pid_s, stride_output_b,         \
        stride_output_h, stride_output_s, offset_output_s_start, offset_output_s_end, half_dim, chunk_size

int skv_apply_rotary_pos_emb_push_cache_opt(const void* x, const void* cos_sin, const int32_t* position_ids,
                                            void* output_cache, const int32_t* cnts, int batch_size, int heads,
                                            int seq_len, int embed_dim, int stride_xb, int stride_xh,
                                            int stride_xs, int stride_xe, int stride_cos_sin, int stride_pid_b,
                                            int stride_pid_h, int stride_pid_s, int stride_output_b,
                                            int stride_output_h, int stride_output_s, int offset_output_s_start,
                                            int offset_output_s_end, int half_dim, int chunk_size,
                                            skv_stream_t stream) {
    if (!x || !cos_sin || !position_ids || !output_cache || !cnts) return SKV_ERR_ARG;
    return finish(skv_launch_rope_chunked(SKV_ROPE_PUSH_ARGS, 0, 1, (hipStream_t)stream));
}

int skv_apply_rotary_pos_emb_push_cache_opt_glm(const void* x, const void* cos_sin, const int32_t* position_ids,
                                                void* output_cache, const int32_t* cnts, int batch_size,
                                                int heads, int seq_len, int embed_dim, int stride_xb,
                                                int stride_xh, int stride_xs, int stride_xe, int stride_cos_sin,
                                                int stride_pid_b, int stride_pid_h, int stride_pid_s,
                                                int stride_output_b, int stride_output_h, int stride_output_s,
                                                int offset_output_s_start, int offset_output_s_end, int half_dim,
                                                int chunk_size, skv_stream_t stream) {
    if (!x || !cos_sin || !position_ids || !output_cache || !cnts) return SKV_ERR_ARG;
    return finish(skv_launch_rope_chunked(SKV_ROPE_PUSH_ARGS, 1, 1, (hipStream_t)stream));
}

int skv_apply_rotary_pos_emb_push_cache(const void* x, const void* cos_sin, const int32_t* position_ids,
                                        void* output_cache, const int32_t* cnts, int batch_size, int heads,
                                        int seq_len, int embed_dim, int stride_xb, int stride_xh, int stride_xs,
                                        int stride_xe, int stride_cos_sin, int stride_pid_b, int stride_pid_h,
                                        int stride_pid_s, int stride_output_b, int stride_output_h,
                                        int stride_output_s, int offset_output_s_start, int offset_output_s_end,
                                        int half_dim, int chunk_size, skv_stream_t stream) {
    if (!x || !cos_sin || !position_ids || !output_cache || !cnts) return SKV_ERR_ARG;
    return finish(skv_launch_rope_chunked(SKV_ROPE_PUSH_ARGS, 0, 1, (hipStream_t)stream));
}

int skv_apply_rotary_pos_emb_new(const void* x, const void* cos_sin, const int64_t* position_ids, void* output,
                                 int batch_size, int heads, int seq_len, int embed_dim, int stride_xb,
                                 int stride_xh, int stride_xs, int stride_xe, int stride_cos_sin,
                                 int stride_pid_b, int stride_pid_h, int stride_pid_s, int half_dim,
                                 skv_stream_t stream) {
    if (!x || !cos_sin || !position_ids || !output) return SKV_ERR_ARG;
    return finish(skv_launch_rope_plain(x, cos_sin, nullptr, 0, position_ids, output, batch_size, heads, seq_len,
                                        embed_dim, stride_xb, stride_xh, stride_xs, stride_xe, stride_cos_sin,
                                        stride_pid_b, stride_pid_h, stride_pid_s, half_dim, (hipStream_t)stream));
}

int skv_apply_rotary_pos_emb_new_v2(const void* x, const void* cos_sin, const int32_t* position_ids, void* output,
                                    int batch_size, int heads, int seq_len, int embed_dim, int stride_xb,
                                    int stride_xh, int stride_xs, int stride_xe, int stride_cos_sin,
                                    int stride_pid_b, int stride_pid_h, int stride_pid_s, int half_dim,
                                    int chunk_size, skv_stream_t stream) {
    if (!x || !cos_sin || !position_ids || !output) return SKV_ERR_ARG;
    return finish(skv_launch_rope_chunked(x, cos_sin, position_ids, output, nullptr, batch_size, heads, seq_len,
                                          embed_dim, stride_xb, stride_xh, stride_xs, stride_xe, stride_cos_sin,
                                          stride_pid_b, stride_pid_h, stride_pid_s, 0, 0, 0, 0, 0, half_dim,
                                          chunk_size, 0, 2, (hipStream_t)stream));
}

int skv_apply_rotary_pos_emb(const void* x, const void* cos, const void* sin, const int64_t* position_ids,
                             void* output, int batch_size, int heads, int seq_len, int embed_dim, int stride_xb,
                             int stride_xh, int stride_xs, int stride_xe, int stride_cos, int stride_sin,
                             int stride_pid_b, int stride_pid_h, int stride_pid_s, int half_dim,
                             skv_stream_t stream) {
    if (!x || !cos || !sin || !position_ids || !output) return SKV_ERR_ARG;
    return finish(skv_launch_rope_plain(x, cos, sin, stride_sin, position_ids, output, batch_size, heads, seq_len,
                                        embed_dim, stride_xb, stride_xh, stride_xs, stride_xe, stride_cos,
                                        stride_pid_b, stride_pid_h, stride_pid_s, half_dim, (hipStream_t)stream));
}

// ---------------------------------------------------------------- part 2: fused decode path
size_t skv_select_workspace_bytes(int blocks, int groups, int n_landmarks) {
    return carve_select_ws(nullptr, blocks, groups, n_landmarks).total;
}

int skv_select_chunks(const void* q, const void* landmarks, const int64_t* landmark_idx, int64_t* cached_pos_ids,
                      int32_t* offsets, int32_t* cnts, void* workspace, void* softmax_out, int64_t* selected_out,
                      int blocks, int groups, int n_landmarks, int select_sets, float alpha, skv_stream_t stream) {
    if (!q || !landmarks || !cached_pos_ids || !offsets || !cnts || !workspace) return SKV_ERR_ARG;
    if (blocks < 1 || n_landmarks < select_sets || select_sets < 1) return SKV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    SelectWs w = carve_select_ws(workspace, blocks, groups, n_landmarks);
    int rc = skv_launch_score(q, landmarks, w.D, w.pmax, w.psum, blocks, groups, n_landmarks, alpha, st);
    if (rc != SKV_OK) return rc;
    rc = skv_launch_normalize_groupmax(w.D, w.pmax, w.psum, softmax_out, w.score, w.score_stride, blocks, groups,
                                       n_landmarks, st);
    if (rc != SKV_OK) return rc;
    return finish(skv_launch_topk_reorder(w.score, w.score_stride, landmark_idx, nullptr, cached_pos_ids, offsets, cnts,
                                          selected_out, nullptr, blocks, n_landmarks, select_sets, st));
}

int skv_select_chunks_inplace(const void* q, const void* landmarks, const int64_t* landmark_idx,
                              int64_t* cached_pos_ids, int32_t* miss_ids, int32_t* dst_slots, int32_t* cnts,
                              void* workspace, void* softmax_out, int64_t* selected_out, int blocks, int groups,
                              int n_landmarks, int select_sets, int resident_sets, int32_t* slot_age, float alpha,
                              skv_stream_t stream) {
    if (!q || !landmarks || !cached_pos_ids || !miss_ids || !dst_slots || !cnts || !workspace) return SKV_ERR_ARG;
    if (blocks < 1 || n_landmarks < select_sets || select_sets < 1 || resident_sets < select_sets) return SKV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    SelectWs w = carve_select_ws(workspace, blocks, groups, n_landmarks);
    int rc = skv_launch_score(q, landmarks, w.D, w.pmax, w.psum, blocks, groups, n_landmarks, alpha, st);
    if (rc != SKV_OK) return rc;
    rc = skv_launch_normalize_groupmax(w.D, w.pmax, w.psum, softmax_out, w.score, w.score_stride, blocks, groups,
                                       n_landmarks, st);
    if (rc != SKV_OK) return rc;
    return finish(skv_launch_topk_resident(w.score, w.score_stride, landmark_idx, nullptr, cached_pos_ids, miss_ids, cnts,
                                           selected_out, dst_slots, blocks, n_landmarks, select_sets, resident_sets,
                                           slot_age, st));
}

// ---- speculative early V fetch (skv_early.hip) -------------------------------------------------------------------
size_t skv_early_state_bytes(int blocks, int groups, int n_landmarks, int n_chunks, int early_max) {
    if (blocks < 1 || groups < 1 || n_landmarks < 1 || n_chunks < 1 || early_max < 1) return 0;
    return skv_carve_early(nullptr, blocks, groups, n_landmarks, n_chunks, early_max).total;
}

static int early_offsets(int blocks, int groups, int n_landmarks, int n_chunks, int early_max, long long* out, int n_out) {
    if (!out || n_out < 0 || blocks < 1 || groups < 1 || n_landmarks < 1 || n_chunks < 1 || early_max < 1) return SKV_ERR_ARG;
    const EarlyState e = skv_carve_early(nullptr, blocks, groups, n_landmarks, n_chunks, early_max);
    const unsigned char* z = nullptr;
    const long long all[SKV_EARLY_STATE_REGIONS] = {
        (const unsigned char*)e.dthr - z,      (const unsigned char*)e.finals - z,    (const unsigned char*)e.flag_cnt - z,
        (const unsigned char*)e.flag_slot - z, (const unsigned char*)e.early_cnt - z, (const unsigned char*)e.early_ids - z,
        (const unsigned char*)e.early_of - z,  (const unsigned char*)e.staging - z,   (const unsigned char*)e.gap_slots - z,
        (const unsigned char*)e.map_ok - z,    (const unsigned char*)e.near_cnt - z,  (const unsigned char*)e.near_ids - z,
        (const unsigned char*)e.near_pub - z};
    for (int i = 0; i < n_out && i < SKV_EARLY_STATE_REGIONS; ++i) out[i] = all[i];
    return SKV_OK;
}

// eight entries, the round-3 contract of this name (a caller built against it passes an 8-entry array)
int skv_early_state_offsets(int blocks, int groups, int n_landmarks, int n_chunks, int early_max, long long* out8) {
    return early_offsets(blocks, groups, n_landmarks, n_chunks, early_max, out8, 8);
}

// the first n_out (<= SKV_EARLY_STATE_REGIONS = 13) entries: 8 = the slot -> chunk-id gap table i32 [B][128], 9 = its validity flag,
// 10 = near-miss count i32 [B], 11 = near-miss list i32 [B][64], 12 = near misses staged now i32 [B][96] (slots E .. E + 95 of 7)
int skv_early_state_offsets2(int blocks, int groups, int n_landmarks, int n_chunks, int early_max, long long* out, int n_out) {
    return early_offsets(blocks, groups, n_landmarks, n_chunks, early_max, out, n_out);
}

int skv_early_state_init(void* state, int blocks, int groups, int n_landmarks, int n_chunks, int early_max,
                         skv_stream_t stream) {
    if (!state || blocks < 1 || groups < 1 || n_landmarks < 1 || n_chunks < 1 || early_max < 1) return SKV_ERR_ARG;
    return finish(skv_launch_early_init(skv_carve_early(state, blocks, groups, n_landmarks, n_chunks, early_max), blocks, groups,
                                        n_landmarks, n_chunks, early_max, (hipStream_t)stream));
}

int skv_early_state_set_landmark_map(void* state, const int64_t* landmark_idx, int blocks, int groups, int n_landmarks,
                                     int n_chunks, int early_max, skv_stream_t stream) {
    if (!state || !landmark_idx || blocks < 1 || groups < 1 || n_landmarks < 1 || n_chunks < 1 || early_max < 1) return SKV_ERR_ARG;
    return finish(skv_launch_early_map(skv_carve_early(state, blocks, groups, n_landmarks, n_chunks, early_max), landmark_idx, blocks,
                                       n_landmarks, (hipStream_t)stream));
}

int skv_select_chunks_inplace_early(const void* q, const void* landmarks, const int64_t* landmark_idx,
                                    int64_t* cached_pos_ids, int32_t* miss_ids, int32_t* dst_slots, int32_t* cnts,
                                    void* workspace, void* softmax_out, int64_t* selected_out, int blocks, int groups,
                                    int n_landmarks, int select_sets, int resident_sets, int32_t* slot_age, float alpha,
                                    void* early_state, const void* v_host, long long host_block_stride, int n_chunks,
                                    int early_max, float margin, skv_stream_t stream) {
    if (!q || !landmarks || !landmark_idx || !cached_pos_ids || !miss_ids || !dst_slots || !cnts || !workspace) return SKV_ERR_ARG;
    if (!early_state || !v_host || (host_block_stride % 8)) return SKV_ERR_ARG;
    if (blocks < 1 || n_landmarks < select_sets || select_sets < 1 || resident_sets < select_sets) return SKV_ERR_ARG;
    if (n_chunks < 1 || early_max < 1 || early_max > 128) return SKV_ERR_ARG;
    // the list (normalise launch) and the pull (second-generation top-k launch) exist for these shapes only; a list that is
    // published MUST be pulled, so other shapes are refused here instead of degrading silently
    if (n_landmarks > 65536 || resident_sets > 1024 || n_chunks > (1 << 18)) return SKV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    SelectWs w = carve_select_ws(workspace, blocks, groups, n_landmarks);
    const EarlyState es = skv_carve_early(early_state, blocks, groups, n_landmarks, n_chunks, early_max);
    EarlyHooks eh = skv_early_hooks(es, blocks, groups, margin, landmark_idx, cached_pos_ids, v_host, host_block_stride, n_landmarks,
                                    resident_sets, n_chunks, early_max);
    int rc = skv_launch_score(q, landmarks, w.D, w.pmax, w.psum, blocks, groups, n_landmarks, alpha, st, &eh);
    if (rc != SKV_OK) return rc;
    rc = skv_launch_normalize_groupmax(w.D, w.pmax, w.psum, softmax_out, w.score, w.score_stride, blocks, groups,
                                       n_landmarks, st, &eh);
    if (rc != SKV_OK) return rc;
    return finish(skv_launch_topk_resident(w.score, w.score_stride, landmark_idx, nullptr, cached_pos_ids, miss_ids, cnts,
                                           selected_out, dst_slots, blocks, n_landmarks, select_sets, resident_sets, slot_age,
                                           st, &eh));
}

// ---- fused selection (round 4): scan (+ keys, slot-major logits) -> top-k with the logit-domain prefilter; no normalise launch
int skv_select_fused_supported(int groups, int n_landmarks, int select_sets) {
    return skv_fused_select_supported(groups, n_landmarks, select_sets) ? 1 : 0;
}

// select state of one layer: log-normalisers f32 [blocks][groups], then the witness levels i32 [blocks], then the last
// launch's diagnostics i32 [blocks][2]
static inline size_t select_state_level_off(int blocks, int groups) { return align256((size_t)blocks * groups * sizeof(float)); }
size_t skv_select_state_stats_offset(int blocks, int groups) {
    return blocks < 1 || groups < 1 ? 0 : select_state_level_off(blocks, groups) + align256((size_t)blocks * sizeof(int));
}
size_t skv_select_state_bytes(int blocks, int groups) {
    return blocks < 1 || groups < 1 ? 0 : skv_select_state_stats_offset(blocks, groups) + align256((size_t)blocks * 2 * sizeof(int));
}

int skv_select_state_init(void* state, int blocks, int groups, skv_stream_t stream) {
    if (!state || blocks < 1 || groups < 1) return SKV_ERR_ARG;
    return hipMemsetAsync(state, 0, skv_select_state_bytes(blocks, groups), (hipStream_t)stream) == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

int skv_select_chunks_fused(const void* q, const void* landmarks, const int64_t* landmark_idx, int64_t* cached_pos_ids,
                            int32_t* miss_ids, int32_t* dst_slots, int32_t* cnts, void* workspace, int64_t* selected_out,
                            int blocks, int groups, int n_landmarks, int select_sets, int resident_sets, int32_t* slot_age,
                            float alpha, void* select_state, void* early_state, const void* v_host, long long host_block_stride,
                            int n_chunks, int early_max, float margin, skv_stream_t stream) {
    if (!q || !landmarks || !cached_pos_ids || !miss_ids || !cnts || !workspace || !select_state) return SKV_ERR_ARG;
    if (blocks < 1 || n_landmarks < select_sets || select_sets < 1 || resident_sets < select_sets) return SKV_ERR_ARG;
    if (!dst_slots && resident_sets != select_sets) return SKV_ERR_ARG;       // a larger resident set: in-place layout only
    if (!skv_fused_select_supported(groups, n_landmarks, select_sets)) return SKV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    SelectWs w = carve_select_ws(workspace, blocks, groups, n_landmarks);
    // the workspace's logit region holds the slot-major logits, its score region the 15-bit keys
    FusedSel fs{(const float*)select_state, (uint16_t*)w.score, w.D, w.score_stride};
    FusedTop ft{w.D, w.pmax, w.psum, (float*)select_state,
                (int*)((unsigned char*)select_state + select_state_level_off(blocks, groups)),
                (int*)((unsigned char*)select_state + skv_select_state_stats_offset(blocks, groups)), (n_landmarks + 255) / 256};
    EarlyHooks eh{};
    const EarlyHooks* hooks = nullptr;
    if (early_state) {
        if (!landmark_idx || !v_host || (host_block_stride % 8) || n_chunks < 1 || early_max < 1 || early_max > 128) return SKV_ERR_ARG;
        if (n_landmarks > 65536 || resident_sets > 1024 || n_chunks > (1 << 18)) return SKV_ERR_UNSUPPORTED;
        const EarlyState es = skv_carve_early(early_state, blocks, groups, n_landmarks, n_chunks, early_max);
        eh = skv_early_hooks(es, blocks, groups, margin, landmark_idx, cached_pos_ids, v_host, host_block_stride, n_landmarks, resident_sets,
                             n_chunks, early_max);
        hooks = &eh;
    }
    int rc = skv_launch_score(q, landmarks, w.D, w.pmax, w.psum, blocks, groups, n_landmarks, alpha, st, hooks, &fs);
    if (rc != SKV_OK) return rc;
    return finish(skv_launch_topk_resident(w.score, w.score_stride, landmark_idx, nullptr, cached_pos_ids, miss_ids, cnts,
                                           selected_out, dst_slots, blocks, n_landmarks, select_sets, resident_sets, slot_age, st,
                                           hooks, &ft, groups));
}

// reference slot order (skv_select_chunks / skv_fetch_kv) with the early fetch
int skv_select_chunks_early(const void* q, const void* landmarks, const int64_t* landmark_idx, int64_t* cached_pos_ids,
                            int32_t* offsets, int32_t* cnts, void* workspace, void* softmax_out, int64_t* selected_out,
                            int blocks, int groups, int n_landmarks, int select_sets, float alpha, void* early_state,
                            const void* v_host, long long host_block_stride, int n_chunks, int early_max, float margin,
                            skv_stream_t stream) {
    if (!q || !landmarks || !landmark_idx || !cached_pos_ids || !offsets || !cnts || !workspace) return SKV_ERR_ARG;
    if (!early_state || !v_host || (host_block_stride % 8)) return SKV_ERR_ARG;
    if (blocks < 1 || n_landmarks < select_sets || select_sets < 1) return SKV_ERR_ARG;
    if (n_chunks < 1 || early_max < 1 || early_max > 128) return SKV_ERR_ARG;
    if (n_landmarks > 65536 || select_sets > 1024 || n_chunks > (1 << 18)) return SKV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    SelectWs w = carve_select_ws(workspace, blocks, groups, n_landmarks);
    const EarlyState es = skv_carve_early(early_state, blocks, groups, n_landmarks, n_chunks, early_max);
    EarlyHooks eh = skv_early_hooks(es, blocks, groups, margin, landmark_idx, cached_pos_ids, v_host, host_block_stride, n_landmarks,
                                    select_sets, n_chunks, early_max);
    int rc = skv_launch_score(q, landmarks, w.D, w.pmax, w.psum, blocks, groups, n_landmarks, alpha, st, &eh);
    if (rc != SKV_OK) return rc;
    rc = skv_launch_normalize_groupmax(w.D, w.pmax, w.psum, softmax_out, w.score, w.score_stride, blocks, groups,
                                       n_landmarks, st, &eh);
    if (rc != SKV_OK) return rc;
    return finish(skv_launch_topk_resident(w.score, w.score_stride, landmark_idx, nullptr, cached_pos_ids, offsets, cnts,
                                           selected_out, nullptr, blocks, n_landmarks, select_sets, select_sets, nullptr, st,
                                           &eh));
}

int skv_fetch_kv_early(const void* U, const void* SV, const void* cos_sin, const int64_t* chunk_ids, const int32_t* cnts,
                       const int32_t* offsets, void* k_cache, const void* k_temp, const void* v_host, void* v_cache,
                       const void* v_temp, int batch_size, int heads, int seq_len, int head_dim, int rank, int select_sets,
                       int chunk_size, long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                       long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                       const void* early_state, int groups, int n_landmarks, int n_chunks, int early_max,
                       skv_stream_t stream) {
    if (!U || !SV || !cos_sin || !chunk_ids || !cnts || !offsets || !k_cache || !k_temp || !v_host || !v_cache || !v_temp)
        return SKV_ERR_ARG;
    if (!early_state || groups < 1 || n_landmarks < 1 || n_chunks < 1 || early_max < 1) return SKV_ERR_ARG;
    if (rope_mode != 1 && rope_mode != 2) return SKV_ERR_ARG;
    const EarlyState es = skv_carve_early((void*)early_state, batch_size * heads, groups, n_landmarks, n_chunks, early_max);
    const EarlyConsume ec{es.early_of, es.staging, n_chunks, early_max + SKV_NEAR_SLOTS};   // (slots [E, E + NEAR): near misses)
    return finish(skv_launch_rebuild(U, SV, cos_sin, chunk_ids, 1, cnts, k_cache, batch_size, heads, seq_len, head_dim,
                                     rank, select_sets, chunk_size, cos_sin_stride, cache_stride_b, cache_stride_h,
                                     cache_stride_s, sparse_start, rope_mode, k_temp, offsets, nullptr, v_host, v_cache, v_temp,
                                     host_block_stride, cache_stride_h, (long long)sparse_start * head_dim,
                                     (hipStream_t)stream, nullptr, &ec));
}

int skv_select_from_scores(const void* scores, int score_stride, const int64_t* landmark_idx, int64_t* cached_pos_ids,
                           int32_t* offsets, int32_t* dst_slots, int32_t* cnts, int64_t* selected_out, int blocks,
                           int n_landmarks, int select_sets, int resident_sets, int32_t* slot_age, skv_stream_t stream) {
    if (!scores || !cached_pos_ids || !offsets || !cnts) return SKV_ERR_ARG;
    if (blocks < 1 || n_landmarks < select_sets || select_sets < 1 || resident_sets < select_sets) return SKV_ERR_ARG;
    return finish(skv_launch_topk_resident(scores, score_stride, landmark_idx, nullptr, cached_pos_ids, offsets, cnts,
                                           selected_out, dst_slots, blocks, n_landmarks, select_sets, resident_sets,
                                           slot_age, (hipStream_t)stream));
}

int skv_score_landmarks(const void* q, const void* landmarks, void* logits, float* part_max, float* part_sum,
                        int blocks, int groups, int n_landmarks, float alpha, skv_stream_t stream) {
    if (!q || !landmarks || !logits || !part_max || !part_sum || blocks < 1 || n_landmarks < 1) return SKV_ERR_ARG;
    return finish(skv_launch_score(q, landmarks, logits, part_max, part_sum, blocks, groups, n_landmarks, alpha,
                                   (hipStream_t)stream));
}

int skv_score_landmarks_early(const void* q, const void* landmarks, const int64_t* landmark_idx, void* logits, float* part_max,
                              float* part_sum, int blocks, int groups, int n_landmarks, float alpha, void* early_state,
                              int n_chunks, int early_max, skv_stream_t stream) {
    if (!q || !landmarks || !landmark_idx || !logits || !part_max || !part_sum || !early_state) return SKV_ERR_ARG;
    if (blocks < 1 || n_landmarks < 1 || n_chunks < 1 || early_max < 1) return SKV_ERR_ARG;
    const EarlyState es = skv_carve_early(early_state, blocks, groups, n_landmarks, n_chunks, early_max);
    EarlyHooks eh{};
    eh.dthr_in = es.dthr; eh.flag_cnt = es.flag_cnt; eh.flag_slot = es.flag_slot; eh.G = groups; eh.lm_idx = landmark_idx;
    eh.T = (n_landmarks + 255) / 256; eh.N = n_landmarks; eh.n_chunks = n_chunks; eh.E = early_max;
    return finish(skv_launch_score(q, landmarks, logits, part_max, part_sum, blocks, groups, n_landmarks, alpha,
                                   (hipStream_t)stream, &eh));
}

/* the scan launch exactly as skv_select_chunks_fused issues it (measurement: bench.py's roofline of the dominant kernel) */
int skv_score_landmarks_fused(const void* q, const void* landmarks, const int64_t* landmark_idx, void* workspace, int blocks,
                              int groups, int n_landmarks, float alpha, void* select_state, void* early_state, int n_chunks,
                              int early_max, skv_stream_t stream) {
    if (!q || !landmarks || !workspace || !select_state || blocks < 1 || n_landmarks < 1) return SKV_ERR_ARG;
    if (!skv_fused_select_supported(groups, n_landmarks, 1)) return SKV_ERR_UNSUPPORTED;
    SelectWs w = carve_select_ws(workspace, blocks, groups, n_landmarks);
    FusedSel fs{(const float*)select_state, (uint16_t*)w.score, w.D, w.score_stride};
    EarlyHooks eh{};
    if (early_state) {
        if (!landmark_idx || n_chunks < 1 || early_max < 1) return SKV_ERR_ARG;
        const EarlyState es = skv_carve_early(early_state, blocks, groups, n_landmarks, n_chunks, early_max);
        eh.dthr_in = es.dthr; eh.flag_cnt = es.flag_cnt; eh.flag_slot = es.flag_slot; eh.G = groups; eh.lm_idx = landmark_idx;
        eh.T = (n_landmarks + 255) / 256; eh.N = n_landmarks; eh.n_chunks = n_chunks; eh.E = early_max;
    }
    return finish(skv_launch_score(q, landmarks, w.D, w.pmax, w.psum, blocks, groups, n_landmarks, alpha, (hipStream_t)stream,
                                   early_state ? &eh : nullptr, &fs));
}

int skv_rebuild_keys(const void* U, const void* SV, const void* cos_sin, const int64_t* chunk_ids,
                     const int32_t* cnts, void* k_cache, int batch_size, int heads, int seq_len, int head_dim,
                     int rank, int select_sets, int chunk_size, long long cos_sin_stride,
                     long long cache_stride_b, long long cache_stride_h, long long cache_stride_s,
                     int sparse_start, int rope_mode, const void* hit_temp, const int32_t* hit_offsets,
                     skv_stream_t stream) {
    if (!U || !SV || !cos_sin || !chunk_ids || !cnts || !k_cache) return SKV_ERR_ARG;
    if (rope_mode != 1 && rope_mode != 2) return SKV_ERR_ARG;
    return finish(skv_launch_rebuild(U, SV, cos_sin, chunk_ids, 1, cnts, k_cache, batch_size, heads, seq_len,
                                     head_dim, rank, select_sets, chunk_size, cos_sin_stride, cache_stride_b,
                                     cache_stride_h, cache_stride_s, sparse_start, rope_mode, hit_temp, hit_offsets,
                                     nullptr, nullptr, nullptr, nullptr, 0, 0, 0, (hipStream_t)stream, nullptr));
}

int skv_fetch_kv(const void* U, const void* SV, const void* cos_sin, const int64_t* chunk_ids, const int32_t* cnts,
                 const int32_t* offsets, void* k_cache, const void* k_temp, const void* v_host, void* v_cache,
                 const void* v_temp, int batch_size, int heads, int seq_len, int head_dim, int rank, int select_sets,
                 int chunk_size, long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                 long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                 skv_stream_t stream) {
    if (!U || !SV || !cos_sin || !chunk_ids || !cnts || !offsets || !k_cache || !k_temp || !v_host || !v_cache || !v_temp)
        return SKV_ERR_ARG;
    if (rope_mode != 1 && rope_mode != 2) return SKV_ERR_ARG;
    return finish(skv_launch_rebuild(U, SV, cos_sin, chunk_ids, 1, cnts, k_cache, batch_size, heads, seq_len, head_dim,
                                     rank, select_sets, chunk_size, cos_sin_stride, cache_stride_b, cache_stride_h,
                                     cache_stride_s, sparse_start, rope_mode, k_temp, offsets, nullptr, v_host, v_cache, v_temp,
                                     host_block_stride, cache_stride_h, (long long)sparse_start * head_dim,
                                     (hipStream_t)stream, nullptr));
}

int skv_fetch_kv_inplace(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                         const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host, void* v_cache,
                         int batch_size, int heads, int seq_len, int head_dim, int rank, int select_sets, int chunk_size,
                         long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                         long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                         skv_stream_t stream) {
    if (!U || !SV || !cos_sin || !miss_ids || !dst_slots || !cnts || !k_cache || !v_host || !v_cache) return SKV_ERR_ARG;
    if (rope_mode != 1 && rope_mode != 2) return SKV_ERR_ARG;
    return finish(skv_launch_rebuild(U, SV, cos_sin, miss_ids, 0, cnts, k_cache, batch_size, heads, seq_len, head_dim,
                                     rank, select_sets, chunk_size, cos_sin_stride, cache_stride_b, cache_stride_h,
                                     cache_stride_s, sparse_start, rope_mode, nullptr, miss_ids, dst_slots, v_host,
                                     v_cache, nullptr, host_block_stride, cache_stride_h,
                                     (long long)sparse_start * head_dim, (hipStream_t)stream, nullptr));
}

int skv_fetch_kv_inplace_early(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                               const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host, void* v_cache,
                               int batch_size, int heads, int seq_len, int head_dim, int rank, int select_sets, int chunk_size,
                               long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                               long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                               const void* early_state, int groups, int n_landmarks, int n_chunks, int early_max,
                               skv_stream_t stream) {
    if (!U || !SV || !cos_sin || !miss_ids || !dst_slots || !cnts || !k_cache || !v_host || !v_cache) return SKV_ERR_ARG;
    if (!early_state || groups < 1 || n_landmarks < 1 || n_chunks < 1 || early_max < 1) return SKV_ERR_ARG;
    if (rope_mode != 1 && rope_mode != 2) return SKV_ERR_ARG;
    const EarlyState es = skv_carve_early((void*)early_state, batch_size * heads, groups, n_landmarks, n_chunks, early_max);
    const EarlyConsume ec{es.early_of, es.staging, n_chunks, early_max + SKV_NEAR_SLOTS};   // (slots [E, E + NEAR): near misses)
    return finish(skv_launch_rebuild(U, SV, cos_sin, miss_ids, 0, cnts, k_cache, batch_size, heads, seq_len, head_dim,
                                     rank, select_sets, chunk_size, cos_sin_stride, cache_stride_b, cache_stride_h,
                                     cache_stride_s, sparse_start, rope_mode, nullptr, miss_ids, dst_slots, v_host,
                                     v_cache, nullptr, host_block_stride, cache_stride_h,
                                     (long long)sparse_start * head_dim, (hipStream_t)stream, nullptr, &ec));
}

static int fetch_kv_attn_inplace_impl(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                              const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host,
                              void* v_cache, const void* q, void* attn_workspace, const int32_t* kv_len_dev, int kv_len,
                              int kv_rows, int batch_size, int heads, int q_heads, int seq_len, int head_dim, int rank, int select_sets,
                              int chunk_size, long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                              long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                              int attn_splits, int resident_sets, float scale, skv_stream_t stream, const short* early_of,
                              const void* early_staging, int early_chunks, int early_max) {
    if (!U || !SV || !cos_sin || !miss_ids || !dst_slots || !cnts || !k_cache || !v_host || !v_cache || !q ||
        !attn_workspace)
        return SKV_ERR_ARG;
    if ((rope_mode != 1 && rope_mode != 2) || heads < 1 || q_heads % heads || chunk_size != 8) return SKV_ERR_ARG;
    if (select_sets % 8) return SKV_ERR_UNSUPPORTED;
    AttnLaunch al{q, attn_workspace, kv_len_dev, kv_len, kv_rows, q_heads / heads, attn_splits, attn_splits + select_sets / 8,
                  scale, resident_sets, early_of, early_staging, early_chunks, early_max};
    return finish(skv_launch_rebuild(U, SV, cos_sin, miss_ids, 0, cnts, k_cache, batch_size, heads, seq_len, head_dim,
                                     rank, select_sets, chunk_size, cos_sin_stride, cache_stride_b, cache_stride_h,
                                     cache_stride_s, sparse_start, rope_mode, nullptr, miss_ids, dst_slots, v_host,
                                     v_cache, nullptr, host_block_stride, cache_stride_h,
                                     (long long)sparse_start * head_dim, (hipStream_t)stream, &al));
}

int skv_fetch_kv_attn_inplace(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                              const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host,
                              void* v_cache, const void* q, void* attn_workspace, const int32_t* kv_len_dev, int kv_len,
                              int kv_rows, int batch_size, int heads, int q_heads, int seq_len, int head_dim, int rank, int select_sets,
                              int chunk_size, long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                              long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                              int attn_splits, int resident_sets, float scale, skv_stream_t stream) {
    return fetch_kv_attn_inplace_impl(U, SV, cos_sin, miss_ids, dst_slots, cnts, k_cache, v_host, v_cache, q, attn_workspace,
                                      kv_len_dev, kv_len, kv_rows, batch_size, heads, q_heads, seq_len, head_dim, rank,
                                      select_sets, chunk_size, cos_sin_stride, cache_stride_b, cache_stride_h, cache_stride_s,
                                      sparse_start, rope_mode, host_block_stride, attn_splits, resident_sets, scale, stream,
                                      nullptr, nullptr, 0, 0);
}

int skv_fetch_kv_attn_inplace_early(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                                    const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host,
                                    void* v_cache, const void* q, void* attn_workspace, const int32_t* kv_len_dev, int kv_len,
                                    int kv_rows, int batch_size, int heads, int q_heads, int seq_len, int head_dim, int rank,
                                    int select_sets, int chunk_size, long long cos_sin_stride, long long cache_stride_b,
                                    long long cache_stride_h, long long cache_stride_s, int sparse_start, int rope_mode,
                                    long long host_block_stride, int attn_splits, int resident_sets, float scale,
                                    const void* early_state, int n_landmarks, int n_chunks, int early_max,
                                    skv_stream_t stream) {
    if (!early_state || heads < 1 || q_heads % heads || n_landmarks < 1 || n_chunks < 1 || early_max < 1) return SKV_ERR_ARG;
    const EarlyState es = skv_carve_early((void*)early_state, batch_size * heads, q_heads / heads, n_landmarks, n_chunks, early_max);
    return fetch_kv_attn_inplace_impl(U, SV, cos_sin, miss_ids, dst_slots, cnts, k_cache, v_host, v_cache, q, attn_workspace,
                                      kv_len_dev, kv_len, kv_rows, batch_size, heads, q_heads, seq_len, head_dim, rank,
                                      select_sets, chunk_size, cos_sin_stride, cache_stride_b, cache_stride_h, cache_stride_s,
                                      sparse_start, rope_mode, host_block_stride, attn_splits, resident_sets, scale, stream,
                                      es.early_of, es.staging, n_chunks, early_max + SKV_NEAR_SLOTS);
}

int skv_attn_finish_inplace(const void* attn_workspace, const int32_t* cnts, void* out, int batch_size, int q_heads,
                            int kv_heads, int select_sets, int attn_splits, skv_stream_t stream) {
    if (!attn_workspace || !cnts || !out) return SKV_ERR_ARG;
    return finish(skv_launch_attn_merge(attn_workspace, cnts, out, batch_size, q_heads, kv_heads, select_sets, attn_splits,
                                        (hipStream_t)stream));
}

int skv_stage_hit_chunks(void* k_cache, void* k_temp, void* v_cache, void* v_temp, const int32_t* offsets,
                         const int32_t* cnts, long long cache_block_stride, long long cache_sparse_offset, int blocks,
                         int select_sets, skv_stream_t stream) {
    if ((!k_cache && !v_cache) || (k_cache && !k_temp) || (v_cache && !v_temp) || !offsets || !cnts) return SKV_ERR_ARG;
    return finish(skv_launch_stage_hits(k_cache, k_temp, v_cache, v_temp, offsets, cnts, cache_block_stride,
                                        cache_sparse_offset, blocks, select_sets, (hipStream_t)stream));
}

int skv_land_chunks(const void* host_values, void* cache_buffer, const void* temp, const int32_t* offsets,
                    const int32_t* cnts, long long host_block_stride, long long cache_block_stride,
                    long long cache_sparse_offset, int blocks, int select_sets, skv_stream_t stream) {
    if (!cache_buffer || !temp || !offsets || !cnts) return SKV_ERR_ARG;
    return finish(skv_launch_land_rows(host_values, cache_buffer, temp, offsets, cnts, host_block_stride,
                                       cache_block_stride, cache_sparse_offset, blocks, select_sets,
                                       (hipStream_t)stream));
}

int skv_sparse_attention(const void* q, const void* k, const void* v, void* out, void* workspace,
                         const int32_t* kv_len_dev, int kv_len, int kv_rows, long long kv_head_stride, int batch_size,
                         int q_heads, int kv_heads, int head_dim, int splits, float scale, skv_stream_t stream) {
    if (!q || !k || !v || !out || !workspace) return SKV_ERR_ARG;
    return finish(skv_launch_sparse_attention(q, k, v, out, workspace, kv_len_dev, kv_len, kv_rows, kv_head_stride,
                                              batch_size, q_heads, kv_heads, head_dim, splits, scale, nullptr, 0, 0, 0,
                                              (hipStream_t)stream));
}

int skv_sparse_attention_slots(const void* q, const void* k, const void* v, void* out, void* workspace,
                               const int32_t* kv_len_dev, int kv_len, int kv_rows, long long kv_head_stride, int batch_size,
                               int q_heads, int kv_heads, int head_dim, int splits, float scale, const int32_t* slots,
                               int select_sets, int sparse_start, int resident_sets, skv_stream_t stream) {
    if (!q || !k || !v || !out || !workspace || !slots || select_sets < 1 || resident_sets < select_sets) return SKV_ERR_ARG;
    return finish(skv_launch_sparse_attention(q, k, v, out, workspace, kv_len_dev, kv_len, kv_rows, kv_head_stride,
                                              batch_size, q_heads, kv_heads, head_dim, splits, scale, slots, select_sets,
                                              sparse_start, resident_sets * 8, (hipStream_t)stream));
}

}  // extern "C"
