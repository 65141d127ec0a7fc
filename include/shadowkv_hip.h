/*
 * shadowkv_hip.h -- C ABI of libshadowkv_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the native half of the ShadowKV decode path.  Part 1 exports one
 * launcher per function of the reference's pybind11 module `kernels.shadowkv`
 * (/root/reference/kernels/main.cu:42-81, prototypes /root/reference/kernels/functions.h:71-463):
 * same argument order and meaning, torch::Tensor replaced by a raw pointer, plus an explicit
 * stream as the last argument (the reference launches on the legacy default stream, except
 * gather_copy_with_offsets which uses the current stream; here the caller always says which).
 * Part 2 exports the fused launchers the decode path actually uses on MI355X (what
 * ShadowKVCache_CPU.get_retrieval_position_ids / get_value_cache / get_key_cache and the
 * attention call of LLM.layer_compute bind to).
 *
 * Conventions: all device pointers; bf16 tensors are `void*`; every tensor contiguous unless
 * strides are passed (strides / lengths in ELEMENTS, as in the reference); `stream` is a
 * hipStream_t (NULL = default stream).  Every function returns 0 on success, <0 on an argument
 * / support error (SKV_ERR_*), and never throws; launches are asynchronous.  Nothing is
 * allocated inside a launcher (graph-capturable), workspaces are passed in.
 */
#ifndef SHADOWKV_HIP_H
#define SHADOWKV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKV_OK 0
#define SKV_ERR_ARG (-1)         /* inconsistent sizes / misaligned strides */
#define SKV_ERR_UNSUPPORTED (-2) /* shape outside what the gfx950 kernels are built for */
#define SKV_ERR_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after the launch */

#if defined(__GNUC__)
#define SKV_EXPORT __attribute__((visibility("default")))
#else
#define SKV_EXPORT
#endif

typedef void* skv_stream_t;

/* library / device info; returns the compiled ABI version (1) */
SKV_EXPORT int skv_abi_version(void);
/* text of the last HIP error seen by a launcher on this thread ("" if none) */
SKV_EXPORT const char* skv_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Part 1: the `kernels.shadowkv` surface (functions.h line cited per entry)
 * ---------------------------------------------------------------------------------------- */

/* functions.h:460  batch_gemm_softmax(A, B, D, Norm, Sum, Softmax, batch_count, m, n, k, alpha, beta)
 * A [batch][m][k], B [batch][n][k], D/Softmax [batch][m][n] bf16; Norm/Sum [batch][ceil(n/256)][m] f32.
 * k must be 128, m in {1,2,4,8,16}.  beta ignored (as in the reference, OnlyAlphaScaling). */
SKV_EXPORT int skv_batch_gemm_softmax(const void* A, const void* B, void* D, float* Norm, float* Sum, void* Softmax,
                           int batch_count, int m, int n, int k, float alpha, float beta, skv_stream_t stream);

/* functions.h:123  reorder_keys_and_compute_offsets(cached_pos_ids, cur_pos_ids, offsets, cnts, bs, heads, map_size)
 * any 1 <= map_size <= 1024 (the reference silently does nothing unless 128/256/512/1024). */
SKV_EXPORT int skv_reorder_keys_and_compute_offsets(int64_t* cached_pos_ids, const int64_t* cur_pos_ids, int32_t* offsets,
                                         int32_t* cnts, int batch_size, int heads, int map_size,
                                         skv_stream_t stream);

/* functions.h:151  gather_copy_with_offsets(values, v_cache_buffer, temp, offsets, cnts, signals, ...)
 * `values` is pinned, device-mapped host memory.  `temp` [batch][heads][map_size][1024] bf16 (the reference's bounce
 * buffer, /root/reference/models/kv_cache.py:612-620) stages the hit rows that move: two launches (stage, land), no
 * inter-workgroup hand-off.  `signals` is accepted and never touched (the reference's 2-CTA flag, copy.cuh:250-263). */
SKV_EXPORT int skv_gather_copy_with_offsets(const void* values, void* v_cache_buffer, void* temp, const int32_t* offsets,
                                 const int32_t* cnts, uint32_t* signals, int batch_size, int heads,
                                 int cpu_v_length, int gpu_v_length, int gpu_v_offset, int gpu_v_stride,
                                 int map_size, skv_stream_t stream);

/* functions.h:97  gather_copy_d2d_with_offsets(keys, offsets, cnts, bs, heads, gpu_k_length, gpu_k_offset,
 * gpu_k_stride, map_size).  Extra argument `temp` (staging, sized as above): the reference serialises a
 * (batch, head) inside one CTA; this library spreads it over map_size/8 workgroups of two launches. */
SKV_EXPORT int skv_gather_copy_d2d_with_offsets(void* keys, const int32_t* offsets, const int32_t* cnts, void* temp,
                                     int batch_size, int heads, int gpu_k_length, int gpu_k_offset,
                                     int gpu_k_stride, int map_size, skv_stream_t stream);

/* functions.h:71  gather_copy(values, v_cache_buffer, position_ids(int64), bs, heads, cpu_v_length, gpu_v_length, map_size) */
SKV_EXPORT int skv_gather_copy(const void* values, void* v_cache_buffer, const int64_t* position_ids, int batch_size,
                    int heads, int cpu_v_length, int gpu_v_length, int map_size, skv_stream_t stream);

/* functions.h:431  batch_gather_gemm(a=U, b=SV, cos, sin, position_ids(int32), output, bs, heads, seq_len,
 * embed_dim, rank, sparse_budget, max_seq_len, chunk_size, offset_array=cnts).  cos / sin are unused by the
 * reference too (its fused epilogue is disabled).  Writes pre-RoPE bf16 rows >= cnt*chunk_size. */
SKV_EXPORT int skv_batch_gather_gemm(const void* a, const void* b, const void* cos, const void* sin,
                          const int32_t* position_ids, void* output, int batch_size, int heads, int seq_len,
                          int embed_dim, int rank, int sparse_budget, int max_seq_len, int chunk_size,
                          const int32_t* offset_array, skv_stream_t stream);

/* functions.h:362 / :396  apply_rotary_pos_emb_push_cache_opt[_glm] and :328 apply_rotary_pos_emb_push_cache
 * (same semantics as _opt). */
SKV_EXPORT int skv_apply_rotary_pos_emb_push_cache_opt(const void* x, const void* cos_sin, const int32_t* position_ids,
                                            void* output_cache, const int32_t* cnts, int batch_size, int heads,
                                            int seq_len, int embed_dim, int stride_xb, int stride_xh,
                                            int stride_xs, int stride_xe, int stride_cos_sin, int stride_pid_b,
                                            int stride_pid_h, int stride_pid_s, int stride_output_b,
                                            int stride_output_h, int stride_output_s, int offset_output_s_start,
                                            int offset_output_s_end, int half_dim, int chunk_size,
                                            skv_stream_t stream);
SKV_EXPORT int skv_apply_rotary_pos_emb_push_cache_opt_glm(const void* x, const void* cos_sin, const int32_t* position_ids,
                                                void* output_cache, const int32_t* cnts, int batch_size,
                                                int heads, int seq_len, int embed_dim, int stride_xb,
                                                int stride_xh, int stride_xs, int stride_xe, int stride_cos_sin,
                                                int stride_pid_b, int stride_pid_h, int stride_pid_s,
                                                int stride_output_b, int stride_output_h, int stride_output_s,
                                                int offset_output_s_start, int offset_output_s_end, int half_dim,
                                                int chunk_size, skv_stream_t stream);
SKV_EXPORT int skv_apply_rotary_pos_emb_push_cache(const void* x, const void* cos_sin, const int32_t* position_ids,
                                        void* output_cache, const int32_t* cnts, int batch_size, int heads,
                                        int seq_len, int embed_dim, int stride_xb, int stride_xh, int stride_xs,
                                        int stride_xe, int stride_cos_sin, int stride_pid_b, int stride_pid_h,
                                        int stride_pid_s, int stride_output_b, int stride_output_h,
                                        int stride_output_s, int offset_output_s_start, int offset_output_s_end,
                                        int half_dim, int chunk_size, skv_stream_t stream);

/* functions.h:240  apply_rotary_pos_emb_new(x, cos_sin, position_ids(int64), output, ...) */
SKV_EXPORT int skv_apply_rotary_pos_emb_new(const void* x, const void* cos_sin, const int64_t* position_ids, void* output,
                                 int batch_size, int heads, int seq_len, int embed_dim, int stride_xb,
                                 int stride_xh, int stride_xs, int stride_xe, int stride_cos_sin,
                                 int stride_pid_b, int stride_pid_h, int stride_pid_s, int half_dim,
                                 skv_stream_t stream);

/* functions.h:281  apply_rotary_pos_emb_new_v2(x, cos_sin, position_ids(int32 chunk ids), output, ..., chunk_size) */
SKV_EXPORT int skv_apply_rotary_pos_emb_new_v2(const void* x, const void* cos_sin, const int32_t* position_ids, void* output,
                                    int batch_size, int heads, int seq_len, int embed_dim, int stride_xb,
                                    int stride_xh, int stride_xs, int stride_xe, int stride_cos_sin,
                                    int stride_pid_b, int stride_pid_h, int stride_pid_s, int half_dim,
                                    int chunk_size, skv_stream_t stream);

/* functions.h:187  apply_rotary_pos_emb(x, cos, sin, position_ids(int64), output, ...) separate full-width tables */
SKV_EXPORT int skv_apply_rotary_pos_emb(const void* x, const void* cos, const void* sin, const int64_t* position_ids,
                             void* output, int batch_size, int heads, int seq_len, int embed_dim, int stride_xb,
                             int stride_xh, int stride_xs, int stride_xe, int stride_cos, int stride_sin,
                             int stride_pid_b, int stride_pid_h, int stride_pid_s, int half_dim,
                             skv_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Part 2: fused decode-path launchers (what the MI355X host code calls)
 * ---------------------------------------------------------------------------------------- */

/* Bytes of scratch skv_select_chunks needs for (blocks = bs*kv_heads, groups, n_landmarks). */
SKV_EXPORT size_t skv_select_workspace_bytes(int blocks, int groups, int n_landmarks);

/* One call = ShadowKVCache_CPU.get_retrieval_position_ids (/root/reference/models/kv_cache.py:983-1057):
 * score q against the landmarks, softmax, max over the GQA group, exact top-`select_sets`
 * (ties -> lowest landmark slot), slot -> chunk id through k_landmark_idx, diff against the
 * resident set.  q [blocks][groups][128] bf16 (q_len == 1), landmarks [blocks][n][128] bf16,
 * landmark_idx int64 [blocks][n], cached_pos_ids int64 [blocks][select_sets] (in/out),
 * offsets int32 [blocks][select_sets], cnts int32 [blocks].  Optional outputs (may be NULL):
 * softmax_out bf16 [blocks][groups][n], selected_out int64 [blocks][select_sets]. */
SKV_EXPORT int skv_select_chunks(const void* q, const void* landmarks, const int64_t* landmark_idx, int64_t* cached_pos_ids,
                      int32_t* offsets, int32_t* cnts, void* workspace, void* softmax_out, int64_t* selected_out,
                      int blocks, int groups, int n_landmarks, int select_sets, float alpha, skv_stream_t stream);

/* The LAST stage of skv_select_chunks[_inplace] alone - what the reference does with torch.topk + gather +
 * reorder_keys_and_compute_offsets (/root/reference/models/kv_cache.py:1031-1055): exact top-`select_sets` of the bf16
 * scores (non-negative: softmax probabilities after the group max; ties at the k-th value -> lowest landmark slot),
 * slot -> chunk id, diff against the resident set.  scores [blocks][score_stride] bf16, score_stride % 8 == 0 and
 * >= n_landmarks, rows 16-B aligned.  dst_slots NULL: reference slot order (offsets = old slot / chunk id as in
 * skv_select_chunks); non-NULL: in-place layout (offsets = miss ids, as in skv_select_chunks_inplace, which also
 * explains resident_sets / slot_age; reference slot order: resident_sets == select_sets, slot_age NULL). */
SKV_EXPORT int skv_select_from_scores(const void* scores, int score_stride, const int64_t* landmark_idx, int64_t* cached_pos_ids,
                           int32_t* offsets, int32_t* dst_slots, int32_t* cnts, int64_t* selected_out, int blocks,
                           int n_landmarks, int select_sets, int resident_sets, int32_t* slot_age, skv_stream_t stream);

/* Stage 1 of skv_select_chunks alone (the HBM-bound landmark scan), for roofline measurement and
 * profiling: logits bf16 [blocks][groups][n] and per-256-landmark partial (max, sum) f32
 * [blocks][ceil(n/256)][groups].  Same kernel, same launch shape as inside skv_select_chunks. */
SKV_EXPORT int skv_score_landmarks(const void* q, const void* landmarks, void* logits, float* part_max, float* part_sum,
                        int blocks, int groups, int n_landmarks, float alpha, skv_stream_t stream);

/* ShadowKVCache_CPU.get_key_cache, rebuild part (kv_cache.py:1157-1168 -> models/tensor_op.py:201-238):
 * k_cache[b][h][sparse_start + i][:] = RoPE(bf16(U[b][pos(i)] . SV[b][h]^T), pos(i)) for chunks >= cnts.
 * chunk_ids int64 [bs][heads][select_sets] (the reordered cached_pos_ids); rope_mode 1 = Llama
 * (cos_sin width 128), 2 = GLM (width 64).  Cache strides in elements.  hit_temp / hit_offsets (nullable):
 * also land the moved hit chunks staged by skv_stage_hit_chunks (the K half of the two-phase compaction). */
SKV_EXPORT int skv_rebuild_keys(const void* U, const void* SV, const void* cos_sin, const int64_t* chunk_ids,
                     const int32_t* cnts, void* k_cache, int batch_size, int heads, int seq_len, int head_dim,
                     int rank, int select_sets, int chunk_size, long long cos_sin_stride,
                     long long cache_stride_b, long long cache_stride_h, long long cache_stride_s,
                     int sparse_start, int rope_mode, const void* hit_temp, const int32_t* hit_offsets,
                     skv_stream_t stream);

/* ShadowKVCache_CPU.get_value_cache (kv_cache.py:1059-1106) / the K-side compaction of get_key_cache (:1140-1150) as the
 * decode path launches them - two phases, no in-kernel synchronisation:
 *   skv_stage_hit_chunks: temp[b][i] <- cache[b][sparse + offsets[i]] for hit chunks whose slot changes, for
 *                         the K and the V buffer in one launch (either may be NULL);
 *   skv_land_chunks     : cache[b][sparse + i] <- temp[b][i] (moved hits) / host_values[b][offsets[i]] (misses;
 *                         host_values NULL: hits only).
 * For K, landing is folded into skv_rebuild_keys (hit_temp / hit_offsets, NULL = not used).  temp buffers are
 * [blocks][select_sets][1024] bf16.  The launch boundary between the two phases is the only ordering needed. */
SKV_EXPORT int skv_stage_hit_chunks(void* k_cache, void* k_temp, void* v_cache, void* v_temp, const int32_t* offsets,
                         const int32_t* cnts, long long cache_block_stride, long long cache_sparse_offset, int blocks,
                         int select_sets, skv_stream_t stream);
SKV_EXPORT int skv_land_chunks(const void* host_values, void* cache_buffer, const void* temp, const int32_t* offsets,
                    const int32_t* cnts, long long host_block_stride, long long cache_block_stride,
                    long long cache_sparse_offset, int blocks, int select_sets, skv_stream_t stream);

/* get_key_cache and get_value_cache of one layer as ONE launch (after skv_stage_hit_chunks): K chunks >= cnts are
 * rebuilt (as skv_rebuild_keys), moved K hits land from k_temp, and the V chunks land from v_temp / the pinned host
 * table (as skv_land_chunks) in extra workgroups of the same grid - no stream fork/join between the two halves.
 * K and V caches must share strides. */
SKV_EXPORT int skv_fetch_kv(const void* U, const void* SV, const void* cos_sin, const int64_t* chunk_ids, const int32_t* cnts,
                 const int32_t* offsets, void* k_cache, const void* k_temp, const void* v_host, void* v_cache,
                 const void* v_temp, int batch_size, int heads, int seq_len, int head_dim, int rank, int select_sets,
                 int chunk_size, long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                 long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                 skv_stream_t stream);

/* Sparse decode attention (replaces flash_attn_with_kvcache at /root/reference/models/base.py:341 for
 * q_len == 1).  q [bs][q_heads][128], k/v [bs][kv_heads][rows][128] (kv_head_stride elements between heads),
 * out [bs][q_heads][128] bf16.  kv_len_dev (int32 on device) overrides kv_len when non-NULL; kv_rows = rows a head
 * owns in k / v: a host kv_len beyond it is SKV_ERR_ARG, a device-side one is clamped to it (the reference's view slice
 * [:sparse_end + gen] clamps the same way, /root/reference/models/kv_cache.py:1100,1172).
 * workspace: skv_attn_workspace_bytes(bs, q_heads, splits). */
SKV_EXPORT size_t skv_attn_workspace_bytes(int batch_size, int q_heads, int splits);
SKV_EXPORT int skv_sparse_attention(const void* q, const void* k, const void* v, void* out, void* workspace,
                         const int32_t* kv_len_dev, int kv_len, int kv_rows, long long kv_head_stride, int batch_size,
                         int q_heads, int kv_heads, int head_dim, int splits, float scale, skv_stream_t stream);

/* skv_sparse_attention over a resident set larger than the selection: of the sparse region [sparse_start, sparse_start +
 * resident_sets * 8) only the chunks in slots[b * kv_heads + h][0 .. select_sets) are attended (the array
 * skv_select_chunks_inplace leaves in dst_slots), rows before and behind the region as usual. */
SKV_EXPORT int skv_sparse_attention_slots(const void* q, const void* k, const void* v, void* out, void* workspace,
                               const int32_t* kv_len_dev, int kv_len, int kv_rows, long long kv_head_stride, int batch_size,
                               int q_heads, int kv_heads, int head_dim, int splits, float scale, const int32_t* slots,
                               int select_sets, int sparse_start, int resident_sets, skv_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Part 3: small fused host-model ops of the decode step (what sits between the dense projections and
 * the ShadowKV kernels in LLM.layer_compute, /root/reference/models/base.py:315-341)
 * ---------------------------------------------------------------------------------------- */

/* Split the fused QKV projection of ONE new token per sequence (qkv [bs][(q_heads+2*kv_heads)*128]), rotate
 * q and k at pos[b] (rope_mode 1 = NeoX / Llama, 2 = GLM), write q [bs][q_heads][128] and push k, v into
 * row *row_idx of the cache buffers [bs][kv_heads][cache_rows][128].  Replaces vllm rotary_embedding
 * (/root/reference/models/llama.py:296) + ShadowKVCache_CPU.update_kv_cache (kv_cache.py:1227-1271); rows
 * outside [0, cache_rows) are dropped like the reference's zero-length slice; pos[b] is clamped to the cos_sin_rows
 * rows of the table.  q_override (nullable): values written to q instead of the rotated projection (synthetic-query
 * benchmarking). */
SKV_EXPORT int skv_qkv_rope_update(const void* qkv, const void* cos_sin, const int64_t* pos, const int64_t* row_idx,
                        const void* q_override, void* q_out, void* k_cache, void* v_cache, int batch_size,
                        int q_heads, int kv_heads, int head_dim, long long cos_sin_stride, int cos_sin_rows,
                        long long cache_stride_b, long long cache_stride_h, int cache_rows, int rope_mode,
                        skv_stream_t stream);

/* h = x + residual (bf16; residual may be NULL), y = RMSNorm(h) * weight (flashinfer.norm.rmsnorm,
 * /root/reference/models/tensor_op.py:34-39).  h_out nullable. */
SKV_EXPORT int skv_add_rmsnorm(const void* x, const void* residual, const void* weight, void* h_out, void* y, int rows,
                    int hidden, float eps, skv_stream_t stream);

/* ShadowKVCache_CPU.update_kv_cache (/root/reference/models/kv_cache.py:1227-1271) as one launch: rows [row0, row0 +
 * incoming) of k_buf / v_buf [bs][heads][buf_rows][128] <- k_new / v_new [bs][heads][incoming][128] (element strides of the
 * new rows passed: the V rows are a transposed view of the fused projection).  Rows at or past buf_rows are dropped, as the
 * reference's zero-length slice does. */
SKV_EXPORT int skv_update_kv_cache(const void* k_new, const void* v_new, void* k_buf, void* v_buf, int batch_size, int heads,
                        int incoming, int head_dim, long long k_stride_b, long long k_stride_h, long long k_stride_s,
                        long long v_stride_b, long long v_stride_h, long long v_stride_s, long long buf_stride_b,
                        long long buf_stride_h, int row0, int buf_rows, skv_stream_t stream);

/* out[r][i] = silu(x[r][i]) * x[r][inter+i]  (vllm._custom_ops.silu_and_mul, llama.py:421) */
SKV_EXPORT int skv_silu_and_mul(const void* x, void* out, int rows, int inter, skv_stream_t stream);

/* y[n] = W[n][:] . x (+ bias[n]) for ONE token: W [N][K] bf16 row-major, x [K], y [N] bf16, f32 accumulation
 * (the decode-time F.linear calls of /root/reference/models/llama.py:380,407,415,424 at q_len == 1, bs == 1).
 * fuse_silu_mul != 0: W = [gate; up] (N = 2I rows), y [I] = silu(gate.x) * (up.x)  (llama.py:415-421).
 * K must be a multiple of 8 and >= 512. */
SKV_EXPORT int skv_gemv_bf16(const void* W, const void* x, const void* bias, void* y, int N, int K, int fuse_silu_mul,
                  skv_stream_t stream);

/* The same projection for M <= 32 token rows (batched decode, bs sequences x q_len 1): X [M][K], Y [M][N] (or [M][N/2]
 * with fuse_silu_mul); the weights stream once, the tokens ride the MFMA N dimension (two 16-token tiles per weight
 * fragment for M > 16).  K % 32 == 0.  Agrees with
 * M calls of skv_gemv_bf16 to f32 accumulation error (different summation order), not bit for bit. */
SKV_EXPORT int skv_linear_rows_bf16(const void* W, const void* X, const void* bias, void* Y, int M, int N, int K,
                         int fuse_silu_mul, skv_stream_t stream);

/* Same with "residual add + RMSNorm" fused in front (K must be 4096): h = x + residual (residual nullable),
 * xn = RMSNorm(h) * norm_weight, y = W . xn (+bias / fused SiLU*mul); h is stored to h_out when non-NULL
 * (the new residual stream).  One launch for tensor_op.layer_norm + F.linear of llama.py:354-380, :410-415. */
SKV_EXPORT int skv_norm_gemv_bf16(const void* W, const void* x, const void* residual, const void* norm_weight, float eps,
                       void* h_out, const void* bias, void* y, int N, int K, int fuse_silu_mul, skv_stream_t stream);

/* [residual add + RMSNorm +] fused QKV projection of ONE token (bs == 1) with the split, RoPE and the cache append
 * in the epilogue (= skv_norm_gemv_bf16 followed by skv_qkv_rope_update, one launch).  norm_weight NULL: x is used
 * as is (then K is arbitrary, multiple of 512); otherwise K must be 4096. */
SKV_EXPORT int skv_qkv_gemv_rope_update(const void* Wqkv, const void* x, const void* residual, const void* norm_weight,
                             float eps, void* h_out, const void* bias, const void* cos_sin, const int64_t* pos,
                             const int64_t* row_idx, const void* q_override, void* q_out, void* k_cache, void* v_cache,
                             int K, int q_heads, int kv_heads, int head_dim, long long cos_sin_stride, int cos_sin_rows,
                             long long cache_stride_h, int cache_rows, int rope_mode, skv_stream_t stream);

/* ---- part 3b: in-place chunk layout (MI355X-first variant of parts 1-2) ---------------------------------------- */

/* skv_select_chunks with an IN-PLACE resident set: chunks selected again keep their slot, the misses (ascending chunk
 * id) take the slots of the evicted chunks (ascending slot).  Same selected SET as skv_select_chunks /
 * /root/reference/models/kv_cache.py:1006-1057 (and the same `selected_out`), a different slot ORDER: no resident row
 * ever moves, so the d2d compaction of reorder_keys_and_compute_offsets + gather_copy_d2d_with_offsets
 * (/root/reference/kernels/map.cuh:754-796, gather_copy.cu) has nothing to do.  Attention is order-independent.
 *   cached_pos_ids [blocks][R] in/out: id per slot (only the freed slots are rewritten; negative = empty slot)
 *   cnts [blocks] out: hits;   for r in [0, S - cnt): miss_ids[b][cnt + r] = id, dst_slots[b][cnt + r] = slot
 *   dst_slots[b][0 .. cnt) = the slots of the hits, ascending (so dst_slots[b][0 .. S) lists the S attended slots)
 * Resident set larger than the selection (MI355X: HBM is plentiful, the PCIe link is the roof): R = resident_sets >=
 * S = select_sets slots per head (R <= 1024).  A selected chunk found in ANY of the R slots is a hit; the S - cnt misses
 * replace the least recently selected of the slots that were not selected in this step (slot_age int32 [blocks][R]
 * in/out: steps since the slot's chunk was last selected, saturating at 62; 63 = empty slot, taken first; ties ->
 * lowest slot).  Attention still runs over exactly the S selected chunks (skv_fetch_kv_attn_inplace,
 * skv_sparse_attention_slots): same outputs, fewer chunks over PCIe.  R == S: the policy above degenerates to "every
 * slot not selected again is replaced" - the reference's resident set; slot_age may be NULL. */
SKV_EXPORT int skv_select_chunks_inplace(const void* q, const void* landmarks, const int64_t* landmark_idx,
                              int64_t* cached_pos_ids, int32_t* miss_ids, int32_t* dst_slots, int32_t* cnts,
                              void* workspace, void* softmax_out, int64_t* selected_out, int blocks, int groups,
                              int n_landmarks, int select_sets, int resident_sets, int32_t* slot_age, float alpha,
                              skv_stream_t stream);

/* skv_fetch_kv for the in-place layout: K rows of the misses rebuilt (U[idx].SV^T + RoPE) and V chunks of the misses
 * fetched from the pinned host table, each written to slot dst_slots[.] of the sparse region; one launch. */
SKV_EXPORT int skv_fetch_kv_inplace(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                         const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host, void* v_cache,
                         int batch_size, int heads, int seq_len, int head_dim, int rank, int select_sets, int chunk_size,
                         long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                         long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                         skv_stream_t stream);

/* skv_fetch_kv_inplace plus the attention, in the same launch: (1) extra workgroups run the split attention pass over
 * the rows that are final before the launch starts (local, outliers, the chunks selected again - dst_slots[.][0 .. cnt)
 * -, generated tokens, which sit behind the resident_sets * 8 rows of the sparse region), so they ride on the CUs the
 * PCIe-bound V fetch leaves idle; (2) every workgroup that builds a miss tile (8 chunks: K rebuilt, V fetched by the same
 * workgroup, host loads issued first) attends its 64 rows from LDS before it exits.  Records go to attn_workspace:
 * skv_attn_workspace_bytes(bs, q_heads, attn_splits + select_sets / 8).  Follow with skv_attn_finish_inplace.  Together
 * they compute exactly what skv_fetch_kv_inplace + skv_sparse_attention compute (flash_attn_with_kvcache at
 * /root/reference/models/base.py:341), with a different order of the f32 sums.  rank 160, chunk_size 8, select_sets % 8
 * == 0, q_heads / heads in {4, 8}; kv_rows = rows per head in the caches (device-side kv_len is clamped to it). */
SKV_EXPORT int skv_fetch_kv_attn_inplace(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                              const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host,
                              void* v_cache, const void* q, void* attn_workspace, const int32_t* kv_len_dev, int kv_len,
                              int kv_rows, int batch_size, int heads, int q_heads, int seq_len, int head_dim, int rank, int select_sets,
                              int chunk_size, long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                              long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                              int attn_splits, int resident_sets, float scale, skv_stream_t stream);

/* ---- fused selection (round 4): two launches instead of three -------------------------------------------------------
 * skv_select_chunks / skv_select_chunks_inplace[_early] without the normalise launch: the scan launch also leaves, per
 * landmark slot, a 15-bit monotone key of max_g (logit_g - c~_g) - the group maximum in the logit domain under the PREVIOUS
 * step's log-normalisers c~_g = m_g + ln s_g, kept in select_state - and the top-k launch (1) computes this step's softmax
 * finals from the tile partials, (2) takes as candidates every slot whose key reaches key(S-th largest kappa - spread of
 * (c_g - c~_g) - 2^-4) - every other slot's score is strictly below S slots' scores, (3) evaluates the candidates' scores
 * EXACTLY (the normalise kernel's arithmetic) and (4) selects the exact top-S among them with the usual tie rule and diff.
 * A step with more than 2,048 candidates (first step, a jump of the query) evaluates every slot in the same launch.  Results are
 * IDENTICAL to the three-launch entries (same selected set, same reordering / miss lists / hit counts).  The reference has no
 * counterpart (three CUTLASS kernels + torch.max / topk / gather, /root/reference/models/kv_cache.py:1006-1042).
 * dst_slots null: the reference's slot order (cached_pos_ids reordered in place, miss_ids = offsets); non-null: in-place layout.
 * select_state: skv_select_state_bytes(blocks, groups) bytes of device memory PER LAYER, zeroed once (skv_select_state_init).
 * Diagnostics of the last launch, int32 [blocks][2] at byte skv_select_state_stats_offset(blocks, groups) of the state: path
 * (0: the carried level held - no search; bit 0: the level was searched; bit 1: every slot was evaluated) and candidate count.
 * early_state (nullable) + v_host .. margin: the speculative early V fetch as in skv_select_chunks_inplace_early; its list role
 * then runs in the pull workgroups of the top-k launch.  groups in {4, 8}, n_landmarks <= 32,768: skv_select_fused_supported. */
SKV_EXPORT int skv_select_fused_supported(int groups, int n_landmarks, int select_sets);
SKV_EXPORT size_t skv_select_state_bytes(int blocks, int groups);
SKV_EXPORT size_t skv_select_state_stats_offset(int blocks, int groups);
SKV_EXPORT int skv_select_state_init(void* state, int blocks, int groups, skv_stream_t stream);
SKV_EXPORT int skv_score_landmarks_fused(const void* q, const void* landmarks, const int64_t* landmark_idx, void* workspace, int blocks,
                              int groups, int n_landmarks, float alpha, void* select_state, void* early_state, int n_chunks,
                              int early_max, skv_stream_t stream);   /* the scan launch of skv_select_chunks_fused alone (measurement) */
SKV_EXPORT int skv_select_chunks_fused(const void* q, const void* landmarks, const int64_t* landmark_idx, int64_t* cached_pos_ids,
                            int32_t* miss_ids, int32_t* dst_slots, int32_t* cnts, void* workspace, int64_t* selected_out,
                            int blocks, int groups, int n_landmarks, int select_sets, int resident_sets, int32_t* slot_age,
                            float alpha, void* select_state, void* early_state, const void* v_host, long long host_block_stride,
                            int n_chunks, int early_max, float margin, skv_stream_t stream);

/* ---- speculative early V fetch (in-place layout) -------------------------------------------------------------------
 * Which chunks a step will MISS is predictable as soon as the scan has produced the logits: a landmark slot whose logit
 * reaches the previous step's k-th value (in logit space, per query head) and whose chunk is not resident.  With an early
 * state, skv_select_chunks_inplace_early flags those slots in the scan launch, turns the flags into a list of up to
 * early_max non-resident chunks per (batch, head) in one extra workgroup of the normalise launch, and pulls them from the
 * pinned host table into an HBM staging buffer in one extra workgroup of the top-k launch - PCIe works while the top-k
 * runs; no extra launch, no extra stream.  skv_fetch_kv_attn_inplace_early then reads every staged miss chunk from HBM and
 * only the rest over PCIe.  Results are IDENTICAL to skv_select_chunks_inplace + skv_fetch_kv_attn_inplace (staged bytes
 * are the host table's bytes; a wrong prediction costs PCIe bytes only, at most early_max chunks per head).  No reference
 * counterpart: the reference fetches after its top-k (/root/reference/models/kv_cache.py:1059-1106).
 * early_state: skv_early_state_bytes(...) bytes of device memory PER LAYER, initialised once with skv_early_state_init;
 * n_chunks = chunks per head of the host table (ids in landmark_idx are < n_chunks); early_max <= 128; margin is added to
 * the logit thresholds (0: flag what would have made the previous top-k; > 0: fewer).  n_landmarks <= 65,536,
 * resident_sets <= 1,024, n_chunks <= 262,144; other shapes: SKV_ERR_UNSUPPORTED (-2), use the plain pair.
 * The two calls are a PAIR: a state whose last step was not consumed by skv_fetch_kv_attn_inplace_early may still be used (the
 * next skv_select_chunks_inplace_early rewrites it), but skv_fetch_kv_attn_inplace_early must only follow the
 * skv_select_chunks_inplace_early of the same step and state.
 * skv_early_state_offsets: byte offsets of the state's regions (diagnostics): 0 thresholds f32 [B][G], 1 finals f32
 * [B][G][2], 2 flag counts i32 [B][T], 3 flagged slots i32 [B][T][16], 4 pulled count i32 [B], 5 pulled chunk ids i32
 * [B][early_max], 6 staging index per chunk i16 [B][n_chunks], 7 staging [B][early_max][2048 B] - EIGHT entries.
 * skv_early_state_offsets2 writes the first n_out (<= SKV_EARLY_STATE_REGIONS) entries: 8, 9 see skv_early_state_set_landmark_map;
 * 10 near-miss counts i32 [2][B], 11 near-miss lists i32 [2][B][64], 12 near misses staged now i32 [2][B][64] (round 5: two lists;
 * region 7, the staging, holds early_max + SKV_NEAR_SLOTS slots per (batch, head); see skv_norm_gemv_near_pull_bf16). */
#ifndef SKV_NEAR_SLOTS
#define SKV_NEAR_SLOTS 128
#endif
#define SKV_EARLY_STATE_REGIONS 13
SKV_EXPORT size_t skv_early_state_bytes(int blocks, int groups, int n_landmarks, int n_chunks, int early_max);
SKV_EXPORT int skv_early_state_offsets(int blocks, int groups, int n_landmarks, int n_chunks, int early_max, long long* out8);
SKV_EXPORT int skv_early_state_offsets2(int blocks, int groups, int n_landmarks, int n_chunks, int early_max, long long* out,
                                        int n_out);
SKV_EXPORT int skv_early_state_init(void* state, int blocks, int groups, int n_landmarks, int n_chunks, int early_max,
                         skv_stream_t stream);
/* Optional, once after skv_early_state_init (round 4): the head's slot -> chunk-id map in closed form.  The reference registers
 * every chunk but the outliers as a landmark, in ascending order (/root/reference/models/kv_cache.py:903-919), so the id of
 * slot j is j + (number of left-out chunks up to there): with the table of up to 127 such gaps the list role needs no dependent
 * gather of landmark_idx (one memory round trip less in front of the link, ~1 us per layer).  A landmark_idx of any other shape
 * (unsorted, ids below their slot, more gaps) is detected here and keeps the gather.  skv_early_state_offsets2 reports them: 8 = the gap
 * table i32 [blocks][128], 9 = its validity flag i32 [blocks]. */
SKV_EXPORT int skv_early_state_set_landmark_map(void* state, const int64_t* landmark_idx, int blocks, int groups, int n_landmarks,
                                     int n_chunks, int early_max, skv_stream_t stream);
SKV_EXPORT int skv_select_chunks_inplace_early(const void* q, const void* landmarks, const int64_t* landmark_idx,
                              int64_t* cached_pos_ids, int32_t* miss_ids, int32_t* dst_slots, int32_t* cnts,
                              void* workspace, void* softmax_out, int64_t* selected_out, int blocks, int groups,
                              int n_landmarks, int select_sets, int resident_sets, int32_t* slot_age, float alpha,
                              void* early_state, const void* v_host, long long host_block_stride, int n_chunks,
                              int early_max, float margin, skv_stream_t stream);
/* skv_score_landmarks as the early selection launches it (the scan with its flag pass; the flags go to the state and are
 * rewritten by the next selection): the kernel bench.py times for the roofline when the early fetch is on. */
SKV_EXPORT int skv_score_landmarks_early(const void* q, const void* landmarks, const int64_t* landmark_idx, void* logits,
                              float* part_max, float* part_sum, int blocks, int groups, int n_landmarks, float alpha,
                              void* early_state, int n_chunks, int early_max, skv_stream_t stream);
/* The same pair for the reference's slot order: skv_select_chunks / skv_fetch_kv with the early state (arguments as there,
 * then the early arguments; `groups` = q_heads / heads).  select_sets <= 256. */
SKV_EXPORT int skv_select_chunks_early(const void* q, const void* landmarks, const int64_t* landmark_idx, int64_t* cached_pos_ids,
                              int32_t* offsets, int32_t* cnts, void* workspace, void* softmax_out, int64_t* selected_out,
                              int blocks, int groups, int n_landmarks, int select_sets, float alpha, void* early_state,
                              const void* v_host, long long host_block_stride, int n_chunks, int early_max, float margin,
                              skv_stream_t stream);
SKV_EXPORT int skv_fetch_kv_early(const void* U, const void* SV, const void* cos_sin, const int64_t* chunk_ids, const int32_t* cnts,
                              const int32_t* offsets, void* k_cache, const void* k_temp, const void* v_host, void* v_cache,
                              const void* v_temp, int batch_size, int heads, int seq_len, int head_dim, int rank, int select_sets,
                              int chunk_size, long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                              long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                              const void* early_state, int groups, int n_landmarks, int n_chunks, int early_max,
                              skv_stream_t stream);
/* skv_fetch_kv_inplace (the plain in-place launch: batches, standalone attention) with the early state. */
SKV_EXPORT int skv_fetch_kv_inplace_early(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                              const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host, void* v_cache,
                              int batch_size, int heads, int seq_len, int head_dim, int rank, int select_sets, int chunk_size,
                              long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                              long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                              const void* early_state, int groups, int n_landmarks, int n_chunks, int early_max,
                              skv_stream_t stream);
SKV_EXPORT int skv_fetch_kv_attn_inplace_early(const void* U, const void* SV, const void* cos_sin, const int32_t* miss_ids,
                              const int32_t* dst_slots, const int32_t* cnts, void* k_cache, const void* v_host,
                              void* v_cache, const void* q, void* attn_workspace, const int32_t* kv_len_dev, int kv_len,
                              int kv_rows, int batch_size, int heads, int q_heads, int seq_len, int head_dim, int rank, int select_sets,
                              int chunk_size, long long cos_sin_stride, long long cache_stride_b, long long cache_stride_h,
                              long long cache_stride_s, int sparse_start, int rope_mode, long long host_block_stride,
                              int attn_splits, int resident_sets, float scale, const void* early_state, int n_landmarks,
                              int n_chunks, int early_max, skv_stream_t stream);

/* Merge of the records skv_fetch_kv_attn_inplace left (resident splits + live miss tiles, told by cnts);
 * out [bs][q_heads][128] bf16. */
SKV_EXPORT int skv_attn_finish_inplace(const void* attn_workspace, const int32_t* cnts, void* out, int batch_size, int q_heads,
                            int kv_heads, int select_sets, int attn_splits, skv_stream_t stream);

/* End of a decode step in one launch: top-p filter over the k <= 64 sorted top-k logits (vals = logits / temperature,
 * descending; idx their token ids), multinomial draw (sample_token, /root/reference/models/tensor_op.py:242-297) written
 * to token[bs], and the device-side step counters advanced: pos[b] += 1; gen += 1 (tokens generated so far); row_idx =
 * base + gen % slack; kv_len = base + min(gen + 1, slack); step_idx = (step_idx + 1) % table_len (step_idx nullable).
 * (gen < slack is the regular case; beyond it the generated rows are a ring of the last `slack` tokens.)
 * Statistics (optional, hit_accum NULL = off): hit_accum[0] += sum of hit_cnts[0 .. n_hit_cnts) - the chunk hit counts the
 * selection kernels of this step left (all layers), so a captured step counts its own hits without extra launches.  Randomness: counter-based hash of
 * (seed, pos[b], b, lane) - reproducible and graph-capturable. */
SKV_EXPORT int skv_sample_advance(const float* vals, const int64_t* idx, int batch_size, int k, float top_p,
                       unsigned long long seed, int64_t* token, int64_t* pos, int64_t* gen, int64_t* row_idx,
                       int32_t* kv_len, int64_t* step_idx, long long base, long long slack, long long table_len,
                       const int32_t* hit_cnts, int n_hit_cnts, int64_t* hit_accum, skv_stream_t stream);

/* skv_sample_advance with the top-k inside: logits bf16 [bs][row_stride] straight from the lm_head (vocab % 8 == 0, vocab
 * <= 524,288, 16-B aligned rows; rows beyond 131,072 logits - GLM-4: 151,552 - are searched in equal parts and merged),
 * exact k-th largest value; every logit above it and EVERY logit tied with it stays (the reference's filter removes logits
 * < the k-th value, /root/reference/models/tensor_op.py:253-255), at most 64 candidates (beyond: lowest token ids of the
 * tied ones); then logit / temperature, top-p, draw and counters as above - one launch for top_k_top_p_filter +
 * multinomial + the step bookkeeping. */
SKV_EXPORT int skv_sample_topk_advance(const void* logits, long long row_stride, int vocab, int batch_size, int k,
                            float temperature, float top_p, unsigned long long seed, int64_t* token, int64_t* pos,
                            int64_t* gen, int64_t* row_idx, int32_t* kv_len, int64_t* step_idx, long long base,
                            long long slack, long long table_len, const int32_t* hit_cnts, int n_hit_cnts,
                            int64_t* hit_accum, skv_stream_t stream);

/* Round 5, near-miss staging ahead of the NEXT decode step: skv_norm_gemv_bf16 with fuse_silu_mul = 1 (the gate/up launch of a
 * layer: residual add + RMSNorm + [gate; up] projection + SiLU * mul, K == 4096) whose first `blocks` x pull_parts workgroups do not compute
 * but stage up to SKV_NEAR_SLOTS chunks per (batch, head) that fell just short of this step's selection - left in the early
 * state by the skv_select_chunks_fused call of the same step and layer - from the pinned host V table into staging slots
 * early_max .. early_max + SKV_NEAR_SLOTS - 1, publishing them in the state's chunk -> staging map.  The *_early fetch launch of
 * the NEXT step reads such a chunk from HBM instead of the host.  Same GEMV result as skv_norm_gemv_bf16, same cache bytes as
 * without it (staged bytes are host-table bytes); no extra launch, no extra stream.  early_state / blocks / groups /
 * n_landmarks / n_chunks / early_max as in skv_early_state_init; v_host / host_block_stride (elements per (batch, head)) as in
 * skv_select_chunks_fused.  pull_parts (1, 2 or 4): pull workgroups per (batch, head) - chunk c belongs to part c % pull_parts,
 * which owns SKV_NEAR_SLOTS / pull_parts of the slots; about 8 workgroups in all is what hides behind the GEMV (8 KV heads: 1). */
SKV_EXPORT int skv_norm_gemv_near_pull_bf16(const void* W, const void* x, const void* residual, const void* norm_weight, float eps,
                                            void* h_out, void* y, int N, int K, void* early_state, int blocks, int groups,
                                            int n_landmarks, int n_chunks, int early_max, const void* v_host,
                                            long long host_block_stride, int pull_parts, int active_lists, skv_stream_t stream);
/* (active_lists: 1 = only this launch stages - the default; 2 = skv_gemv_near_pull_bf16 stages list 1 as well: each role then keeps a
 * chunk either list wants and does not pull what the other one holds.)
 * The same role for near-miss list `list` (0: the 64 candidates just below the selection, what the call above stages; 1: the next
 * 64, staging slots early_max + 64 ..) in a plain one-token GEMV launch with N <= 8192 rows - skv_gemv_bf16, the layer's down
 * projection (bias: the residual riding in the bias slot, or NULL).  Same GEMV result. */
SKV_EXPORT int skv_gemv_near_pull_bf16(const void* W, const void* x, const void* bias, void* y, int N, int K, void* early_state,
                                       int blocks, int groups, int n_landmarks, int n_chunks, int early_max, const void* v_host,
                                       long long host_block_stride, int pull_parts, int list, skv_stream_t stream);
/* The lm_head and the sampler without streaming the logit row through one CU (round 4): skv_norm_gemv_rangemax_bf16 is
 * skv_norm_gemv_bf16 (no fused SiLU) that ALSO leaves, per 16 consecutive outputs, the largest one as an order-preserving
 * 16-bit key (bf16 x >= 0: x | 0x8000; x < 0: ~x) in range_max[N / 16] (N % 16 == 0); skv_sample_topk_advance_ranges is
 * skv_sample_topk_advance that finds the ranges able to hold a top-k logit from those keys (the k-th largest range maximum
 * bounds the k-th largest logit from below) and reads only them - same winners, same draw, bit for bit (a row with more
 * than 128 qualifying ranges - thousands of tied logits - is streamed as before).  vocab % 16 == 0, vocab <= 262,144,
 * range_max 16-B aligned, range_stride (keys between rows) % 8 == 0.  The reference has no counterpart (torch.topk over
 * the f32 logits, /root/reference/models/tensor_op.py:242-297). */
SKV_EXPORT int skv_norm_gemv_rangemax_bf16(const void* W, const void* x, const void* residual, const void* norm_weight, float eps,
                                void* h_out, const void* bias, void* y, int N, int K, void* range_max, skv_stream_t stream);
SKV_EXPORT int skv_sample_topk_advance_ranges(const void* logits, long long row_stride, int vocab, const void* range_max,
                                   long long range_stride, int batch_size, int k, float temperature, float top_p,
                                   unsigned long long seed, int64_t* token, int64_t* pos, int64_t* gen, int64_t* row_idx,
                                   int32_t* kv_len, int64_t* step_idx, long long base, long long slack, long long table_len,
                                   const int32_t* hit_cnts, int n_hit_cnts, int64_t* hit_accum, skv_stream_t stream);

/* Page-locked, device-mapped host memory of EXACTLY nbytes for the chunked V table (hipHostMalloc through the HIP runtime
 * this library is linked against - the one the caller's streams and tensors come from).  The reference pins V with
 * torch.zeros(..., pin_memory=True) (/root/reference/models/kv_cache.py:554-563); torch's pinned allocator rounds 8.19 GB
 * up to 16 GiB, so the cache allocates the table here.  Not launchers: they allocate / free and must not be captured. */
SKV_EXPORT int skv_host_alloc(void** out, size_t nbytes);
SKV_EXPORT int skv_host_free(void* p);

/* ---- part 4: prefill-side state builder ---------------------------------------- */

/* Chunk means (landmark candidates) and per-chunk minimum cosine similarity (outlier score) of the post-RoPE keys,
 * one pass over K: /root/reference/models/kv_cache.py:854 (key_states_roped_ctx.mean(dim=-2)) and :859-868
 * (cosine_similarity(...).min(dim=-1).values), with torch's bf16 rounding points.
 * k [blocks][rows >= chunks*chunk_size][head_dim] bf16, block_stride in elements; means [blocks][chunks][head_dim]
 * bf16, min_cos [blocks][chunks] bf16.  chunk_size must be 8, head_dim 128. */
SKV_EXPORT int skv_chunk_stats(const void* k, long long block_stride, int blocks, int chunks, int chunk_size,
                    int head_dim, void* means, void* min_cos, skv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SHADOWKV_HIP_H */
