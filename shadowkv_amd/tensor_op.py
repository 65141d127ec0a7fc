"""Tensor-op surface the reference's host models import (`/root/reference/models/tensor_op.py`),
re-implemented on PyTorch-ROCm ops + libshadowkv_hip.so.

Names, argument order and return values follow the reference so `models/llama.py` /
`models/base.py`-style host code can import from here unchanged:
  layer_norm, apply_rotary_pos_emb, apply_rotary_pos_emb_single, apply_rotary_pos_emb_cuda,
  apply_rotary_pos_emb_cuda_push_cache, batch_gather_gemm_rotary_pos_emb_cuda, sample_token, ...
plus what replaces third-party CUDA packages on this path:
  sparse_attention_decode  (flash_attn_with_kvcache at base.py:341, q_len == 1)
  silu_and_mul, rotary_embedding_neox  (vllm._custom_ops at llama.py:296,421)
`minference_prefill_kernel` (MInference sparse prefill) is outside the decode path and raises.
"""
import math

import torch
import torch.nn.functional as F

from ._lib import lib, check, ptr, current_stream_handle
from .kernels import shadowkv


# ---------------------------------------------------------------------------- norms / activations
def layer_norm(hidden_states, eps, w):
    """RMSNorm (reference: flashinfer.norm.rmsnorm, tensor_op.py:34-39; the commented-out torch
    version at :41-51 is the spec: f32 variance, normalise, cast back, scale by w)."""
    shape = hidden_states.shape
    out = F.rms_norm(hidden_states.reshape(-1, shape[-1]), (shape[-1],), w, eps)
    return out.view(shape)


def silu_and_mul(out, x):
    """out = silu(x[..., :d]) * x[..., d:]  (vllm._custom_ops.silu_and_mul, llama.py:421)."""
    d = x.shape[-1] // 2
    torch.mul(F.silu(x[..., :d]), x[..., d:], out=out)
    return out


MAX_GEMV_ROWS = 32   # token rows served by the native kernels (1: GEMV, 2..32: small-M MFMA kernel); more -> F.linear


def linear_decode(x, w, bias=None, fuse_silu_mul=False, out=None, near_pull=None):
    """F.linear for decode-time activations (x [..., K] with a handful of token rows): one token -> the native GEMV,
    2..MAX_GEMV_ROWS tokens -> the native small-M MFMA kernel (weights stream once); falls back to F.linear for more
    rows or shapes the kernels are not built for.  fuse_silu_mul: w = [gate; up] -> silu(gate.x) * (up.x).
    near_pull (the down projection of a ShadowKV layer, one token, N <= 8192; ShadowKVCache_CPU.near_pull_args(layer, 1)): the
    launch's first workgroups stage the second near-miss list of this step's selection (skv_gemv_near_pull_bf16); same y."""
    K = x.shape[-1]
    rows = x.numel() // K
    if near_pull is not None:
        N = w.shape[0]
        if rows != 1 or fuse_silu_mul or N > 8192 or K % 32 or K < 512 or not x.is_contiguous() or not w.is_contiguous():
            raise ValueError("near_pull rides in a one-token GEMV launch with at most 8,192 output rows")
        y = out if out is not None else torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        check(lib().skv_gemv_near_pull_bf16(ptr(w), ptr(x), ptr(bias), ptr(y), N, K, *near_pull, current_stream_handle()),
              "gemv_near_pull")
        return y
    if rows > MAX_GEMV_ROWS or K % 32 or K < 512 or not x.is_contiguous() or not w.is_contiguous():
        out = F.linear(x, w, bias)
        return silu_and_mul_fused(out) if fuse_silu_mul else out
    N = w.shape[0]
    No = (N // 2) if fuse_silu_mul else N
    y = out if out is not None else torch.empty(x.shape[:-1] + (No,), dtype=x.dtype, device=x.device)
    if rows == 1:
        check(lib().skv_gemv_bf16(ptr(w), ptr(x), ptr(bias), ptr(y), N, K, 1 if fuse_silu_mul else 0,
                                  current_stream_handle()), "gemv")
    else:
        check(lib().skv_linear_rows_bf16(ptr(w), ptr(x), ptr(bias), ptr(y), rows, N, K, 1 if fuse_silu_mul else 0,
                                         current_stream_handle()), "linear_rows")
    return y


def range_max_supported(x_numel, K, N):
    """Whether norm_linear_decode(..., range_max=) takes the one-launch path that fills the sampler's range keys."""
    return x_numel == K == 4096 and N % 16 == 0 and N // 16 <= 16384


def norm_linear_decode(x, residual, norm_w, eps, w, bias=None, fuse_silu_mul=False, out=None, h_out=None, range_max=None,
                       near_pull=None):
    """(h, y): h = x + residual (residual None -> h = x), y = linear(RMSNorm(h) * norm_w) for one token, one
    native launch when the hidden size is 4096; otherwise add_rmsnorm + linear_decode.  out / h_out: caller-owned
    result buffers (the eager decode paths reuse theirs instead of allocating per call).
    range_max (lm_head only; int16 [>= N // 16], 16-B aligned; range_max_supported(...) must hold): the launch also leaves the
    largest of every 16 outputs as the sampler's 16-bit key (sample_token_native(..., range_max=)).
    near_pull (gate/up launch of a ShadowKV layer only; ShadowKVCache_CPU.near_pull_args(layer)): the launch's first workgroups
    stage the near misses of this step's selection ahead of the next step (skv_norm_gemv_near_pull_bf16); same y."""
    K = x.shape[-1]
    N = w.shape[0]
    if near_pull is not None:
        if not fuse_silu_mul or bias is not None or range_max is not None or x.numel() != K or K != 4096 \
                or not x.is_contiguous() or not w.is_contiguous():
            raise ValueError("near_pull rides in the one-token gate/up launch (hidden size 4096, fused SiLU * mul, no bias)")
        y = out if out is not None else torch.empty(x.shape[:-1] + (N // 2,), dtype=x.dtype, device=x.device)
        h = (h_out if h_out is not None else torch.empty_like(x)) if residual is not None else x
        check(lib().skv_norm_gemv_near_pull_bf16(ptr(w), ptr(x), ptr(residual), ptr(norm_w), float(eps),
                                                 ptr(h) if residual is not None else 0, ptr(y), N, K, *near_pull,
                                                 current_stream_handle()), "norm_gemv_near_pull")
        return h, y
    if range_max is not None:
        if fuse_silu_mul or not range_max_supported(x.numel(), K, N) or not x.is_contiguous() or not w.is_contiguous():
            raise ValueError("range_max: one token, hidden size 4096, N % 16 == 0, N <= 262,144, no fused SiLU")
        y = out if out is not None else torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        h = (h_out if h_out is not None else torch.empty_like(x)) if residual is not None else x
        check(lib().skv_norm_gemv_rangemax_bf16(ptr(w), ptr(x), ptr(residual), ptr(norm_w), float(eps),
                                                ptr(h) if residual is not None else 0, ptr(bias), ptr(y), N, K, ptr(range_max),
                                                current_stream_handle()), "norm_gemv_rangemax")
        return h, y
    if x.numel() != K or K != 4096 or not x.is_contiguous() or not w.is_contiguous():
        h, hs = add_rmsnorm(x, residual, norm_w, eps)
        return h, linear_decode(hs, w, bias, fuse_silu_mul, out=out)
    y = out if out is not None else torch.empty(x.shape[:-1] + ((N // 2) if fuse_silu_mul else N,), dtype=x.dtype, device=x.device)
    h = (h_out if h_out is not None else torch.empty_like(x)) if residual is not None else x
    check(lib().skv_norm_gemv_bf16(ptr(w), ptr(x), ptr(residual), ptr(norm_w), float(eps),
                                   ptr(h) if residual is not None else 0, ptr(bias), ptr(y), N, K,
                                   1 if fuse_silu_mul else 0, current_stream_handle()), "norm_gemv")
    return h, y


def add_rmsnorm(x, residual, w, eps):
    """(h, y): h = x + residual (bf16; residual None -> h = x), y = RMSNorm(h) * w.  One native launch."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    h = torch.empty_like(x2) if residual is not None else x2
    y = torch.empty_like(x2)
    check(lib().skv_add_rmsnorm(ptr(x2), ptr(residual.reshape(-1, shape[-1]) if residual is not None else None),
                                ptr(w), ptr(h) if residual is not None else 0, ptr(y), x2.shape[0], shape[-1],
                                float(eps), current_stream_handle()), "add_rmsnorm")
    return h.view(shape), y.view(shape)


def silu_and_mul_fused(x):
    """silu(x[..., :d]) * x[..., d:] in one native launch (x contiguous)."""
    d = x.shape[-1] // 2
    out = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
    check(lib().skv_silu_and_mul(ptr(x), ptr(out), x.numel() // x.shape[-1], d, current_stream_handle()),
          "silu_and_mul")
    return out


def qkv_rope_update(qkv, cos_sin, pos, row_idx, k_cache, v_cache, q_heads, kv_heads, q_override=None, q_out=None):
    """qkv [bs, 1, (Hq+2Hkv)*D] -> q [bs, Hq, 1, D] rotated at pos [bs] (int64, device); k (rotated) and v are
    written into row row_idx[0] (int64, device) of k_cache / v_cache [bs, Hkv, rows, D].  One native launch."""
    bs = qkv.shape[0]
    D = k_cache.shape[-1]
    width = cos_sin.shape[-1]
    q = q_out if q_out is not None else torch.empty(bs, q_heads, 1, D, dtype=qkv.dtype, device=qkv.device)
    check(lib().skv_qkv_rope_update(ptr(qkv), ptr(cos_sin), ptr(pos), ptr(row_idx), ptr(q_override), ptr(q),
                                    ptr(k_cache), ptr(v_cache), bs, q_heads, kv_heads, D, cos_sin.stride(0),
                                    cos_sin.shape[0], k_cache.stride(0), k_cache.stride(1), k_cache.shape[2],
                                    1 if width == 128 else 2,
                                    current_stream_handle()), "qkv_rope_update")
    return q


def norm_qkv_rope_update(x, residual, norm_w, eps, wqkv, bqkv, cos_sin, pos, row_idx, k_cache, v_cache, q_heads,
                         kv_heads, q_override=None):
    """(h, q) for ONE token of ONE sequence: h = x + residual, q = RoPE(split(linear(RMSNorm(h)))) [1, Hq, 1, D];
    rotated k and v are pushed into row row_idx of k_cache / v_cache [1, Hkv, rows, D].  One native launch when the
    hidden size is 4096 and bs == 1; otherwise norm_linear_decode + qkv_rope_update."""
    K = x.shape[-1]
    D = k_cache.shape[-1]
    if x.numel() != K or K != 4096 or not x.is_contiguous() or not wqkv.is_contiguous() or D != 128:
        h, qkv = norm_linear_decode(x, residual, norm_w, eps, wqkv, bqkv)
        return h, qkv_rope_update(qkv, cos_sin, pos, row_idx, k_cache, v_cache, q_heads, kv_heads, q_override)
    q = torch.empty(1, q_heads, 1, D, dtype=x.dtype, device=x.device)
    h = torch.empty_like(x) if residual is not None else x
    check(lib().skv_qkv_gemv_rope_update(ptr(wqkv), ptr(x), ptr(residual), ptr(norm_w), float(eps),
                                         ptr(h) if residual is not None else 0, ptr(bqkv), ptr(cos_sin), ptr(pos),
                                         ptr(row_idx), ptr(q_override), ptr(q), ptr(k_cache), ptr(v_cache), K, q_heads,
                                         kv_heads, D, cos_sin.stride(0), cos_sin.shape[0], k_cache.stride(1),
                                         k_cache.shape[2], 1 if cos_sin.shape[-1] == 128 else 2, current_stream_handle()),
          "qkv_gemv_rope_update")
    return h, q


# ---------------------------------------------------------------------------- RoPE (pure torch, tensor_op.py:127-151)
def rotate_half(x):
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def apply_rotary_pos_emb(q, k, cos, sin, position_ids):
    cos = cos[position_ids].unsqueeze(1)
    sin = sin[position_ids].unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def apply_rotary_pos_emb_single(q, cos, sin, position_ids, unsqueeze_dim=1):
    if position_ids.dim() == 3:  # [bs, heads, seq] -> one position per (b, h, s)
        flat = position_ids.reshape(-1, position_ids.size(-1))
        return (q * cos[flat]) + (rotate_half(q) * sin[flat])
    cos = cos[position_ids].unsqueeze(unsqueeze_dim)
    sin = sin[position_ids].unsqueeze(unsqueeze_dim)
    return (q * cos) + (rotate_half(q) * sin)


# ---------------------------------------------------------------------------- RoPE (native, tensor_op.py:154-238)
def apply_rotary_pos_emb_cuda(x, cos_sin, position_ids):
    bs, heads, seq_len, dim = x.shape
    out = torch.empty_like(x)
    shadowkv.apply_rotary_pos_emb_new(x, cos_sin, position_ids, out, bs, heads, seq_len, dim,
                                      x.stride(0), x.stride(1), x.stride(2), x.stride(3), cos_sin.stride(0),
                                      position_ids.stride(0), position_ids.stride(1), position_ids.stride(2),
                                      dim // 2)
    return out


def apply_rotary_pos_emb_cuda_push_cache(x, cos_sin, position_ids, chunk_size, cache, sparse_start, sparse_end, cnts):
    bs, heads, seq_len, dim = x.shape
    width = cos_sin.shape[-1]
    if width == 128:
        fn = shadowkv.apply_rotary_pos_emb_push_cache_opt
    elif width == 64:
        fn = shadowkv.apply_rotary_pos_emb_push_cache_opt_glm
    else:
        raise ValueError(f"Invalid cos_sin shape {cos_sin.shape}")
    fn(x, cos_sin, position_ids, cache, cnts, bs, heads, seq_len, dim,
       x.stride(0), x.stride(1), x.stride(2), x.stride(3), cos_sin.stride(0),
       position_ids.stride(0), position_ids.stride(1), position_ids.stride(2),
       cache.stride(0), cache.stride(1), cache.stride(2), int(sparse_start), int(sparse_end), dim // 2,
       int(chunk_size))
    return cache


def batch_gather_gemm_rotary_pos_emb_cuda(a, b, cos_sin, position_ids, output, chunk_size, cache, sparse_start,
                                          sparse_end, cnts):
    """Two-launch form with the reference's signature (gather-GEMM into `output`, then RoPE-and-push).
    The decode path of ShadowKVCache_CPU uses the fused single launch (`rebuild_keys`) instead."""
    bs, seq_len, rank = a.shape
    _, heads, head_dim, _ = b.shape
    max_seq_len = cos_sin.shape[0]
    num_chunks = position_ids.shape[-1]
    pid32 = position_ids.to(torch.int32).contiguous()
    shadowkv.batch_gather_gemm(a.contiguous(), b.contiguous(), cos_sin, cos_sin, pid32, output, bs, heads, seq_len,
                               head_dim, rank, num_chunks * chunk_size, max_seq_len, chunk_size, cnts)
    return apply_rotary_pos_emb_cuda_push_cache(output, cos_sin, pid32, chunk_size, cache, sparse_start, sparse_end,
                                                cnts)


def rebuild_keys(U, SV, cos_sin, position_ids, cnts, cache, sparse_start, chunk_size, hit_temp=None,
                 hit_offsets=None):
    """Fused K rebuild: cache[b,h,sparse_start+i] = RoPE(bf16(U[b,pos(i)].SV[b,h]^T)) for chunks >= cnts.
    U [bs, seq, r], SV [bs, heads, 128, r], position_ids int64 [bs, heads, S], cache [bs, heads, rows, 128]."""
    bs, seq_len, rank = U.shape
    heads, head_dim = SV.shape[1], SV.shape[2]
    width = cos_sin.shape[-1]
    if width not in (128, 64):
        raise ValueError(f"Invalid cos_sin shape {cos_sin.shape}")
    check(lib().skv_rebuild_keys(ptr(U), ptr(SV), ptr(cos_sin), ptr(position_ids), ptr(cnts), ptr(cache), bs, heads,
                                 seq_len, head_dim, rank, position_ids.shape[-1], int(chunk_size), cos_sin.stride(0),
                                 cache.stride(0), cache.stride(1), cache.stride(2), int(sparse_start),
                                 1 if width == 128 else 2, ptr(hit_temp), ptr(hit_offsets), current_stream_handle()),
          "rebuild_keys")
    return cache


# ---------------------------------------------------------------------------- attention
_attn_ws = {}


def attention_workspace(device, bs, Hq, splits):
    """Per (device, shape, stream) scratch for the split records of the attention kernels (streams may run
    concurrently, so they never share one)."""
    key = (device.index, bs, Hq, splits, current_stream_handle())
    ws = _attn_ws.get(key)
    if ws is None:
        ws = torch.empty(lib().skv_attn_workspace_bytes(bs, Hq, splits), dtype=torch.uint8, device=device)
        _attn_ws[key] = ws
    return ws


def default_attention_splits(bs, Hkv, rows):
    """Splits of the standalone attention pass when the caller names none: one workgroup per (kv head, split) to fill the
    256 CUs (profiles/r02_attn_mfma_probe.txt: 64 splits beat 32 for 4 KV heads), at most 60 (the combine kernel merges up
    to 62 records); long rows (full-attention baseline: 125 K keys per head) get a split per ~2 K keys whatever the batch,
    so that every CU holds several workgroups."""
    return max(1, min(60, max(256 // max(1, bs * Hkv), -(-rows // 2048))))


def sparse_attention_decode(q, k_cache, v_cache, kv_len=None, kv_len_dev=None, splits=None, out=None, slots=None,
                            select_sets=0, sparse_start=0, resident_sets=0):
    """softmax(q K^T / sqrt(D)) V for q_len == 1 over the first kv_len rows of the cache views.
    q [bs, Hq, 1, D] or [bs, Hq, D]; k_cache / v_cache [bs, Hkv, rows, D] views of contiguous
    [bs, Hkv, buf_rows, D] buffers (what get_key_cache / get_value_cache return).  Returns
    [bs, 1, Hq, D] like flash_attn_with_kvcache's output for q [bs, 1, Hq, D].
    slots (int32 [bs * Hkv, select_sets], with sparse_start / resident_sets): a resident set larger than the selection -
    of the rows [sparse_start, sparse_start + 8 * resident_sets) only the listed chunks are attended."""
    bs, Hq = q.shape[0], q.shape[1]
    D = q.shape[-1]
    Hkv = k_cache.shape[1]
    rows = k_cache.shape[2]
    if kv_len is None:
        kv_len = rows
    if kv_len_dev is None and not 1 <= int(kv_len) <= rows:
        raise ValueError(f"kv_len {kv_len} outside the {rows} rows of the cache view")
    if k_cache.stride(3) != 1 or k_cache.stride(2) != D or k_cache.stride(0) != Hkv * k_cache.stride(1) \
            or k_cache.stride() != v_cache.stride():
        raise ValueError("k/v cache views must be row-contiguous slices of [bs, Hkv, rows, D] buffers")
    if not q.is_contiguous():
        q = q.contiguous()
    if splits is None:
        splits = default_attention_splits(bs, Hkv, rows)
    ws = attention_workspace(q.device, bs, Hq, splits)
    if out is None:
        out = torch.empty(bs, 1, Hq, D, dtype=q.dtype, device=q.device)
    if slots is not None:
        check(lib().skv_sparse_attention_slots(ptr(q), ptr(k_cache), ptr(v_cache), ptr(out), ptr(ws), ptr(kv_len_dev),
                                               int(kv_len), rows, k_cache.stride(1), bs, Hq, Hkv, D, splits,
                                               1.0 / math.sqrt(D), ptr(slots), int(select_sets), int(sparse_start),
                                               int(resident_sets), current_stream_handle()), "sparse_attention_slots")
        return out
    check(lib().skv_sparse_attention(ptr(q), ptr(k_cache), ptr(v_cache), ptr(out), ptr(ws), ptr(kv_len_dev),
                                     int(kv_len), rows, k_cache.stride(1), bs, Hq, Hkv, D, splits, 1.0 / math.sqrt(D),
                                     current_stream_handle()), "sparse_attention")
    return out


def chunk_stats(k_ctx, chunk_size=8):
    """k_ctx bf16 [bs, kv, rows, D] (row-contiguous view, rows >= chunks*chunk_size used from row 0), D = 128 ->
    (means [bs, kv, chunks, D], min_cos [bs, kv, chunks]): kv_cache.py:854-868 in one native pass over K."""
    bs, kv, rows, D = k_ctx.shape
    if k_ctx.stride(3) != 1 or k_ctx.stride(2) != D or k_ctx.stride(0) != kv * k_ctx.stride(1):
        raise ValueError("k_ctx must be a row-contiguous [bs, kv, rows, D] view")
    chunks = rows // chunk_size
    means = torch.empty(bs, kv, chunks, D, dtype=k_ctx.dtype, device=k_ctx.device)
    min_cos = torch.empty(bs, kv, chunks, dtype=k_ctx.dtype, device=k_ctx.device)
    check(lib().skv_chunk_stats(ptr(k_ctx), k_ctx.stride(1), bs * kv, chunks, chunk_size, D, ptr(means), ptr(min_cos),
                                current_stream_handle()), "chunk_stats")
    return means, min_cos


def minference_prefill_kernel(*args, **kwargs):
    raise NotImplementedError("MInference sparse prefill is outside the decode hot path (SURVEY.md section 2a)")


# ---------------------------------------------------------------------------- sampling (tensor_op.py:242-297)
def top_k_top_p_filter(logits, top_k=0, top_p=0.0):
    if top_k > 0:
        kth = torch.topk(logits, min(top_k, logits.size(-1)))[0][:, [-1]]
        logits[logits < kth] = float("-inf")
    if top_p > 0.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=True)
        remove = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1) > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        logits[remove.scatter(1, sorted_idx, remove)] = float("-inf")
    return logits


def norm_logits(logits, temperature=0.6, top_k=-1, top_p=0.9):
    assert logits.dim() == 2
    if temperature != 1.0:
        logits = logits / temperature
    return F.softmax(top_k_top_p_filter(logits, top_k=top_k, top_p=top_p), dim=-1)


def sample(probs, num_samples=1):
    return torch.multinomial(probs, num_samples=num_samples, replacement=True)


_sampler_state = {}
_sampler_seed = None


def set_sampler_seed(seed):
    """Seed of the native sampler's counter-based random numbers (None: follow torch.initial_seed(), i.e. what
    torch.manual_seed(...) last set).  A new seed restarts every draw counter, so `manual_seed(s)` followed by the same calls
    draws the same tokens."""
    global _sampler_seed
    _sampler_seed = None if seed is None else int(seed)


def sample_token_native(logits, temperature, top_k, top_p, seed=None, state=None, range_max=None):
    """sample_token in ONE native launch for bf16 logits [bs, V] on the GPU (skv_sample_topk_advance: exact k-th value,
    every logit tied with it kept - the reference's filter -, logit / temperature, nucleus, draw).  Same distribution as
    the torch pipeline below; the random numbers are NOT torch's generator stream: they come from a counter-based hash of
    (seed, draw counter, row) with seed = `seed`, else set_sampler_seed(), else torch.initial_seed() (so torch.manual_seed
    reseeds it), and a draw counter kept per `state` (any dict a caller owns - one per model or generation - default: one
    per (device, batch size) of the process).  Returns None when the kernel does not take the row (caller falls back).
    range_max (int16 [bs, >= V // 16], from norm_linear_decode(..., range_max=) of the SAME logits): the sampler reads the
    keys and the ~k ranges that can hold a winner instead of the whole row - same token."""
    bs, V = logits.shape
    k = min(top_k, V) if top_k > 0 else V
    if not (logits.is_cuda and logits.dtype == torch.bfloat16 and temperature > 0.0 and 1 <= k <= 64 and V % 8 == 0
            and V <= 4 * 131072 and logits.stride(-1) == 1 and logits.stride(0) % 8 == 0 and logits.data_ptr() % 16 == 0):
        return None
    if seed is None:
        seed = _sampler_seed if _sampler_seed is not None else torch.initial_seed()
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = (logits.device.index, bs)
    holder = _sampler_state if state is None else state
    st = holder.get(key)
    if st is None:
        dev = logits.device
        st = dict(pos=torch.zeros(bs, 1, dtype=torch.long, device=dev), gen=torch.zeros(1, dtype=torch.long, device=dev),
                  row=torch.zeros(1, dtype=torch.long, device=dev), kvl=torch.zeros(1, dtype=torch.int32, device=dev),
                  seed=seed)
        holder[key] = st
    elif st["seed"] != seed:            # reseeded: the draw counter starts again
        st["gen"].zero_()
        st["pos"].zero_()
        st["seed"] = seed
    token = torch.empty(bs, 1, dtype=torch.long, device=logits.device)
    if range_max is not None and V % 16 == 0 and V // 16 <= 16384 and V // 16 >= k:
        rm = range_max.view(bs, -1)
        check(lib().skv_sample_topk_advance_ranges(ptr(logits), logits.stride(0), V, ptr(rm), rm.stride(0), bs, k, float(temperature),
                                                   float(top_p), seed, ptr(token), ptr(st["pos"]), ptr(st["gen"]), ptr(st["row"]),
                                                   ptr(st["kvl"]), 0, 0, 1, 1, 0, 0, 0, current_stream_handle()),
              "sample_topk_advance_ranges")
        return token
    check(lib().skv_sample_topk_advance(ptr(logits), logits.stride(0), V, bs, k, float(temperature), float(top_p), seed,
                                        ptr(token), ptr(st["pos"]), ptr(st["gen"]), ptr(st["row"]), ptr(st["kvl"]), 0, 0, 1, 1,
                                        0, 0, 0, current_stream_handle()), "sample_topk_advance")
    return token


def sample_token(logits, temperature=0, top_k=50, top_p=0.9, state=None, range_max=None):
    """tensor_op.py:291-297.  bf16 logits on the GPU take the native sampler, whose random numbers are not torch's generator
    stream (see sample_token_native: seeded from torch.initial_seed() / set_sampler_seed, draw counter per `state`)."""
    if temperature == 0.0:
        return logits.argmax(dim=-1, keepdim=True)
    if logits.dtype == torch.bfloat16 and logits.is_cuda:     # the lm_head's own output: one native launch
        tok = sample_token_native(logits, temperature, top_k, top_p, state=state, range_max=range_max)
        if tok is not None:
            return tok
        logits = logits.float()
    return sample(norm_logits(logits, temperature=temperature, top_p=top_p, top_k=top_k))
