"""`kernels.shadowkv`-compatible module backed by libshadowkv_hip.so (gfx950).

Same twelve function names, argument order and in-place semantics as the reference's pybind11
extension (/root/reference/kernels/main.cu:42-81; prototypes /root/reference/kernels/functions.h).
Tensors are borrowed for the launch only; every launch goes to torch's CURRENT stream (the
reference uses the legacy default stream for all but gather_copy_with_offsets; with an explicit
stream the caller's `with torch.cuda.stream(...)` is always honoured).

dtype / layout errors raise (the reference throws c10::Error from data_ptr<T>()); shapes the
gfx950 kernels are not built for raise ShadowKVNativeError instead of silently doing nothing
(the reference's map_size dispatch is a silent no-op outside 128/256/512/1024).
"""
import torch

from .._lib import lib, check, ptr, current_stream_handle

_staging = {}


def _bf16(t, name, device_ok=("cuda",)):
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name}: expected bfloat16, got {t.dtype}")
    return t


def _dt(t, dtype, name):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _d2d_staging(device, rows):
    """Staging rows (2 KiB each) for the hit chunks gather_copy_d2d_with_offsets moves; the reference's signature has
    no bounce buffer for the K side (one CTA per (batch, head) serialises it, copy.cuh:649-687).  Per (device, stream):
    two streams may compact concurrently.  Allocated at the first call (or by reserve_d2d_staging); never under stream
    capture - a captured launch sequence must not allocate: reserve before capturing."""
    key = (device.type, device.index, current_stream_handle())
    s = _staging.get(key)
    if s is None or s.shape[0] < rows:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("gather_copy_d2d_with_offsets needs a staging buffer of %d rows and the stream is being "
                               "captured: call shadowkv.reserve_d2d_staging(device, rows) under this stream first" % rows)
        s = torch.empty(rows, 1024, dtype=torch.bfloat16, device=device)
        _staging[key] = s
    return s


def reserve_d2d_staging(device, rows):
    """Allocates the staging buffer of gather_copy_d2d_with_offsets for the current stream ahead of time (rows =
    batch_size * heads * map_size)."""
    return _d2d_staging(torch.device(device), rows)


def release_d2d_staging():
    """Drops every staging buffer (they are kept per (device, stream) until released)."""
    _staging.clear()


def gather_copy(values, v_cache_buffer, position_ids, batch_size, heads, cpu_v_length, gpu_v_length, map_size):
    """functions.h:71 -- no-cache gather of `map_size` chunk rows per (batch, head) from pinned host memory."""
    _bf16(values, "values"); _bf16(v_cache_buffer, "v_cache_buffer"); _dt(position_ids, torch.int64, "position_ids")
    check(lib().skv_gather_copy(ptr(values), ptr(v_cache_buffer), ptr(position_ids), batch_size, heads,
                                cpu_v_length, gpu_v_length, map_size, current_stream_handle()), "gather_copy")


def gather_copy_d2d_with_offsets(keys, offsets, cnts, batch_size, heads, gpu_k_length, gpu_k_offset,
                                 gpu_k_stride, map_size):
    """functions.h:97 -- in-place compaction of the hit rows of the key cache's sparse region."""
    _bf16(keys, "keys"); _dt(offsets, torch.int32, "offsets"); _dt(cnts, torch.int32, "cnts")
    tmp = _d2d_staging(keys.device, batch_size * heads * map_size)
    check(lib().skv_gather_copy_d2d_with_offsets(ptr(keys), ptr(offsets), ptr(cnts), ptr(tmp), batch_size, heads,
                                                 gpu_k_length, gpu_k_offset, gpu_k_stride, map_size,
                                                 current_stream_handle()), "gather_copy_d2d_with_offsets")


def reorder_keys_and_compute_offsets(cached_pos_ids, cur_pos_ids, offsets, cnts, batch_size, heads, map_size):
    """functions.h:123 -- hit/miss diff of the selected chunk ids against the resident ones."""
    _dt(cached_pos_ids, torch.int64, "cached_pos_ids"); _dt(cur_pos_ids, torch.int64, "cur_pos_ids")
    _dt(offsets, torch.int32, "offsets"); _dt(cnts, torch.int32, "cnts")
    check(lib().skv_reorder_keys_and_compute_offsets(ptr(cached_pos_ids), ptr(cur_pos_ids), ptr(offsets),
                                                     ptr(cnts), batch_size, heads, map_size,
                                                     current_stream_handle()), "reorder_keys_and_compute_offsets")


def gather_copy_with_offsets(values, v_cache_buffer, temp, offsets, cnts, signals, batch_size, heads,
                             cpu_v_length, gpu_v_length, gpu_v_offset, gpu_v_stride, map_size):
    """functions.h:151 -- hit rows compacted in place, miss rows fetched from pinned host memory.  `temp` must hold
    batch_size * heads * map_size rows of 1024 bf16 (the reference's size, kv_cache.py:612-620)."""
    _bf16(values, "values"); _bf16(v_cache_buffer, "v_cache_buffer"); _bf16(temp, "temp")
    if temp.numel() < batch_size * heads * map_size * 1024:
        raise ValueError(f"temp: {temp.numel()} elements, need {batch_size * heads * map_size * 1024}")
    _dt(offsets, torch.int32, "offsets"); _dt(cnts, torch.int32, "cnts"); _dt(signals, torch.int32, "signals")
    check(lib().skv_gather_copy_with_offsets(ptr(values), ptr(v_cache_buffer), ptr(temp), ptr(offsets), ptr(cnts),
                                             ptr(signals), batch_size, heads, cpu_v_length, gpu_v_length,
                                             gpu_v_offset, gpu_v_stride, map_size, current_stream_handle()),
          "gather_copy_with_offsets")


def apply_rotary_pos_emb(x, cos, sin, position_ids, output, batch_size, heads, seq_len, embed_dim,
                         stride_xb, stride_xh, stride_xs, stride_xe, stride_cos, stride_sin,
                         stride_pid_b, stride_pid_h, stride_pid_s, half_dim):
    """functions.h:187 -- NeoX RoPE with separate full-width cos / sin tables."""
    _bf16(x, "x"); _bf16(cos, "cos"); _bf16(sin, "sin"); _dt(position_ids, torch.int64, "position_ids")
    check(lib().skv_apply_rotary_pos_emb(ptr(x), ptr(cos), ptr(sin), ptr(position_ids), ptr(output), batch_size,
                                         heads, seq_len, embed_dim, stride_xb, stride_xh, stride_xs, stride_xe,
                                         stride_cos, stride_sin, stride_pid_b, stride_pid_h, stride_pid_s, half_dim,
                                         current_stream_handle()), "apply_rotary_pos_emb")


def apply_rotary_pos_emb_new(x, cos_sin, position_ids, output, batch_size, heads, seq_len, embed_dim,
                             stride_xb, stride_xh, stride_xs, stride_xe, stride_cos_sin,
                             stride_pid_b, stride_pid_h, stride_pid_s, half_dim):
    """functions.h:240 -- NeoX RoPE, fused cos|sin table, int64 position per (b, h, s)."""
    _bf16(x, "x"); _bf16(cos_sin, "cos_sin"); _dt(position_ids, torch.int64, "position_ids")
    check(lib().skv_apply_rotary_pos_emb_new(ptr(x), ptr(cos_sin), ptr(position_ids), ptr(output), batch_size,
                                             heads, seq_len, embed_dim, stride_xb, stride_xh, stride_xs, stride_xe,
                                             stride_cos_sin, stride_pid_b, stride_pid_h, stride_pid_s, half_dim,
                                             current_stream_handle()), "apply_rotary_pos_emb_new")


def apply_rotary_pos_emb_new_v2(x, cos_sin, position_ids, output, batch_size, heads, seq_len, embed_dim,
                                stride_xb, stride_xh, stride_xs, stride_xe, stride_cos_sin,
                                stride_pid_b, stride_pid_h, stride_pid_s, half_dim, chunk_size):
    """functions.h:281 -- NeoX RoPE addressed by int32 chunk ids."""
    _bf16(x, "x"); _bf16(cos_sin, "cos_sin"); _dt(position_ids, torch.int32, "position_ids")
    check(lib().skv_apply_rotary_pos_emb_new_v2(ptr(x), ptr(cos_sin), ptr(position_ids), ptr(output), batch_size,
                                                heads, seq_len, embed_dim, stride_xb, stride_xh, stride_xs,
                                                stride_xe, stride_cos_sin, stride_pid_b, stride_pid_h,
                                                stride_pid_s, half_dim, chunk_size, current_stream_handle()),
          "apply_rotary_pos_emb_new_v2")


def _push(fn, name, x, cos_sin, position_ids, output_cache, cnts, *ints):
    _bf16(x, "x"); _bf16(cos_sin, "cos_sin"); _bf16(output_cache, "output_cache")
    _dt(position_ids, torch.int32, "position_ids"); _dt(cnts, torch.int32, "cnts")
    if len(ints) != 19:
        raise TypeError(f"{name}: expected 19 integer arguments, got {len(ints)}")
    check(fn(ptr(x), ptr(cos_sin), ptr(position_ids), ptr(output_cache), ptr(cnts), *[int(i) for i in ints],
             current_stream_handle()), name)


def apply_rotary_pos_emb_push_cache(x, cos_sin, position_ids, output_cache, cnts, *ints):
    """functions.h:328"""
    _push(lib().skv_apply_rotary_pos_emb_push_cache, "apply_rotary_pos_emb_push_cache", x, cos_sin, position_ids,
          output_cache, cnts, *ints)


def apply_rotary_pos_emb_push_cache_opt(x, cos_sin, position_ids, output_cache, cnts, *ints):
    """functions.h:362 -- RoPE rows of chunks >= cnts and push them into the key cache (Llama)."""
    _push(lib().skv_apply_rotary_pos_emb_push_cache_opt, "apply_rotary_pos_emb_push_cache_opt", x, cos_sin,
          position_ids, output_cache, cnts, *ints)


def apply_rotary_pos_emb_push_cache_opt_glm(x, cos_sin, position_ids, output_cache, cnts, *ints):
    """functions.h:396 -- GLM interleaved half-dim variant."""
    _push(lib().skv_apply_rotary_pos_emb_push_cache_opt_glm, "apply_rotary_pos_emb_push_cache_opt_glm", x, cos_sin,
          position_ids, output_cache, cnts, *ints)


def batch_gather_gemm(a, b, cos, sin, position_ids, output, batch_size, heads, seq_len, embed_dim, rank,
                      sparse_budget, max_seq_len, chunk_size, offset_array):
    """functions.h:431 -- output[b,h,i,:] = bf16(U[b, pos(i)] . SV[b,h]^T) for chunks >= offset_array."""
    _bf16(a, "a"); _bf16(b, "b"); _bf16(output, "output")
    _dt(position_ids, torch.int32, "position_ids"); _dt(offset_array, torch.int32, "offset_array")
    check(lib().skv_batch_gather_gemm(ptr(a), ptr(b), ptr(cos), ptr(sin), ptr(position_ids), ptr(output),
                                      batch_size, heads, seq_len, embed_dim, rank, sparse_budget, max_seq_len,
                                      chunk_size, ptr(offset_array), current_stream_handle()), "batch_gather_gemm")


def batch_gemm_softmax(A, B, D, Norm, Sum, Softmax, batch_count, m, n, k, alpha=1.0, beta=0.0):
    """functions.h:460 -- D = bf16(alpha*A.B^T), Softmax = row softmax of D (bf16), partials in Norm / Sum."""
    _bf16(A, "A"); _bf16(B, "B"); _bf16(D, "D"); _bf16(Softmax, "Softmax")
    _dt(Norm, torch.float32, "Norm"); _dt(Sum, torch.float32, "Sum")
    check(lib().skv_batch_gemm_softmax(ptr(A), ptr(B), ptr(D), ptr(Norm), ptr(Sum), ptr(Softmax), batch_count, m, n,
                                       k, float(alpha), float(beta), current_stream_handle()), "batch_gemm_softmax")
