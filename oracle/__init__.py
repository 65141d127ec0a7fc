"""ctypes binding of the CPU oracle (oracle/shadowkv_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under shadowkv_amd/ may import this.

All functions take CPU torch tensors (bf16 / int32 / int64 / f32, contiguous)
and mirror the argument lists of the reference's `kernels.shadowkv` functions
(/root/reference/kernels/functions.h) so tests can call oracle and device code
with the same arguments.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libshadowkv_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "shadowkv_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.device.type == "cpu", "oracle works on CPU tensors only"
    assert t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


_i = ctypes.c_int
_l = ctypes.c_long
_f = ctypes.c_float


def batch_gemm_softmax(A, B, D, Norm, Sum, Softmax, batch_count, m, n, k, alpha, beta=0.0):
    lib().oracle_batch_gemm_softmax(_p(A), _p(B), _p(D), _p(Norm), _p(Sum), _p(Softmax),
                                    _i(batch_count), _i(m), _i(n), _i(k), _f(alpha), _f(beta))


def num_threads():
    """OpenMP threads the oracle's parallel loops run on."""
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(_i(int(n)))


def bind_threads(cpus, whole_set=False):
    """Binds OpenMP thread t of the current team to cpus[t % len(cpus)] (see oracle_bind_threads; thread 0 is the caller);
    whole_set: every thread may run on all of `cpus` again."""
    arr = (ctypes.c_int * len(cpus))(*cpus)
    return int(lib().oracle_bind_threads(arr, len(cpus), _i(1 if whole_set else 0)))


def thread_cpus():
    out = (ctypes.c_int * 1024)(*([-1] * 1024))
    n = int(lib().oracle_thread_cpus(out, 1024))
    return list(out)[:n]


def first_touch_clone(t):
    """A copy of CPU tensor t whose pages are first written by the OpenMP threads (static partition) instead of the caller."""
    assert t.device.type == "cpu" and t.is_contiguous()
    out = torch.empty_like(t)
    lib().oracle_parallel_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib().oracle_parallel_copy(out.data_ptr(), t.data_ptr(), t.numel() * t.element_size())
    return out


def group_max_topk(P, landmark_idx, blocks, groups, n, topk):
    """P bf16 [blocks, groups, n] -> int64 [blocks, topk] (ascending slot order)."""
    out = torch.empty(blocks, topk, dtype=torch.int64)
    rc = lib().oracle_group_max_topk(_p(P), _p(landmark_idx), _p(out), _i(blocks), _i(groups), _i(n), _i(topk))
    if rc != 0:
        raise ValueError(f"topk={topk} exceeds number of landmarks n={n}")
    return out


def fused_candidates(D, fin_m, fin_inv, ctil, topk, level=None):
    """Candidate rule of the device's fused selection (oracle_fused_candidates): D bf16 [blocks, groups, n] logits, fin_m /
    fin_inv f32 [blocks, groups] softmax finals, ctil f32 [blocks, groups], level int32 [blocks] (optional witness levels; a
    level fewer than topk keys reach is ignored) -> (mask uint8 [blocks, n], counts int32 [blocks])."""
    blocks, groups, n = D.shape
    mask = torch.empty(blocks, n, dtype=torch.uint8)
    counts = torch.empty(blocks, dtype=torch.int32)
    lib().oracle_fused_candidates(_p(D), _p(fin_m.contiguous()), _p(fin_inv.contiguous()), _p(ctil.contiguous()), _i(blocks),
                                  _i(groups), _i(n), _i(topk), _p(None if level is None else level.contiguous()), _p(mask), _p(counts))
    return mask, counts


def reorder_keys_and_compute_offsets(cached_pos_ids, cur_pos_ids, offsets, cnts, batch_size, heads, map_size):
    assert cached_pos_ids.dtype == torch.int64 and cur_pos_ids.dtype == torch.int64
    assert offsets.dtype == torch.int32 and cnts.dtype == torch.int32
    lib().oracle_reorder_keys_and_compute_offsets(_p(cached_pos_ids), _p(cur_pos_ids), _p(offsets), _p(cnts),
                                                  _i(batch_size), _i(heads), _i(map_size))


def gather_copy_d2d_with_offsets(keys, offsets, cnts, batch_size, heads, gpu_k_length, gpu_k_offset,
                                 gpu_k_stride, map_size):
    lib().oracle_gather_copy_d2d_with_offsets(_p(keys), _p(offsets), _p(cnts), _i(batch_size), _i(heads),
                                              _i(gpu_k_length), _i(gpu_k_offset), _i(gpu_k_stride), _i(map_size))


def gather_copy_with_offsets(values, v_cache_buffer, temp, offsets, cnts, signals, batch_size, heads,
                             cpu_v_length, gpu_v_length, gpu_v_offset, gpu_v_stride, map_size):
    lib().oracle_gather_copy_with_offsets(_p(values), _p(v_cache_buffer), _p(offsets), _p(cnts),
                                          _i(batch_size), _i(heads), _i(cpu_v_length), _i(gpu_v_length),
                                          _i(gpu_v_offset), _i(gpu_v_stride), _i(map_size))


def gather_copy(values, v_cache_buffer, position_ids, batch_size, heads, cpu_v_length, gpu_v_length, map_size):
    lib().oracle_gather_copy(_p(values), _p(v_cache_buffer), _p(position_ids), _i(batch_size), _i(heads),
                             _i(cpu_v_length), _i(gpu_v_length), _i(map_size))


def batch_gather_gemm(a, b, cos, sin, position_ids, output, batch_size, heads, seq_len, embed_dim, rank,
                      sparse_budget, max_seq_len, chunk_size, offset_array):
    assert position_ids.dtype == torch.int32
    lib().oracle_batch_gather_gemm(_p(a), _p(b), _p(position_ids), _p(output), _i(batch_size), _i(heads),
                                   _i(seq_len), _i(embed_dim), _i(rank), _i(sparse_budget), _i(chunk_size),
                                   _p(offset_array))


def _rope_push(x, cos_sin, position_ids, output_cache, cnts, batch_size, heads, seq_len, embed_dim,
               sxb, sxh, sxs, sxe, scs, spb, sph, sps, sob, soh, sos, off_start, off_end, half_dim,
               chunk_size, glm):
    assert sxe == 1 and position_ids.dtype == torch.int32
    lib().oracle_rope_push_cache(_p(x), _p(cos_sin), _p(position_ids), _p(output_cache), _p(cnts),
                                 _i(batch_size), _i(heads), _i(seq_len), _i(embed_dim), _l(sxb), _l(sxh),
                                 _l(sxs), _l(scs), _l(spb), _l(sph), _l(sps), _l(sob), _l(soh), _l(sos),
                                 _i(off_start), _i(off_end), _i(half_dim), _i(chunk_size), _i(glm))


def apply_rotary_pos_emb_push_cache_opt(*args):
    _rope_push(*args, 0)


def apply_rotary_pos_emb_push_cache_opt_glm(*args):
    _rope_push(*args, 1)


def apply_rotary_pos_emb_new(x, cos_sin, position_ids, output, batch_size, heads, seq_len, embed_dim,
                             sxb, sxh, sxs, sxe, scs, spb, sph, sps, half_dim):
    assert sxe == 1 and position_ids.dtype == torch.int64
    lib().oracle_rope_new(_p(x), _p(cos_sin), _p(position_ids), _p(output), _i(batch_size), _i(heads),
                          _i(seq_len), _i(embed_dim), _l(sxb), _l(sxh), _l(sxs), _l(scs), _l(spb), _l(sph),
                          _l(sps), _i(half_dim))


def sparse_attention(q, k, v, kv_len, scale):
    """q [bs, q_heads, D] bf16; k, v [bs, kv_heads, rows, D] bf16 -> (out bf16, out f32)."""
    bs, qh, d = q.shape
    kvh, rows = k.shape[1], k.shape[2]
    out = torch.empty(bs, qh, d, dtype=torch.bfloat16)
    out32 = torch.empty(bs, qh, d, dtype=torch.float32)
    lib().oracle_sparse_attention(_p(q), _p(k), _p(v), _p(out), _p(out32), _i(bs), _i(qh), _i(kvh), _i(d),
                                  _i(kv_len), _l(rows), _f(scale))
    return out, out32


def sparse_attention_p16(q, k, v, kv_len, scale, grp, ord_, with_flip=False):
    """sparse_attention with the softmax weights rounded to bf16 where the device kernels round them (flash-attn, the
    reference's attention, casts P to bf16 before P.V: models/base.py:341).  grp / ord_ int32 [bs, kv_heads, kv_len]: rounding
    group of every row (-1: never rounded) and its step inside the group (see oracle_sparse_attention_p16).
    -> (out bf16, out f32[, flip f32: the largest |weight x v| of a rounded row per output - one flipped bf16 rounding of a
    weight moves the output by at most 2^-7 times it])."""
    bs, qh, d = q.shape
    kvh, rows = k.shape[1], k.shape[2]
    assert grp.dtype == torch.int32 and ord_.dtype == torch.int32 and grp.shape == (bs, kvh, kv_len) == ord_.shape
    out = torch.empty(bs, qh, d, dtype=torch.bfloat16)
    out32 = torch.empty(bs, qh, d, dtype=torch.float32)
    flip = torch.empty(bs, qh, d, dtype=torch.float32) if with_flip else None
    lib().oracle_sparse_attention_p16(_p(q), _p(k), _p(v), _p(out), _p(out32), _i(bs), _i(qh), _i(kvh), _i(d),
                                      _i(kv_len), _l(rows), _f(scale), _p(grp), _p(ord_), _p(flip))
    return (out, out32, flip) if with_flip else (out, out32)


def chunk_stats(k_ctx):
    """k_ctx bf16 [blocks, rows, 128] (rows = chunks * 8) -> (means bf16 [blocks, chunks, 128], min_cos bf16
    [blocks, chunks]): the chunk means and the per-chunk minimum cosine similarity of kv_cache.py:854-868."""
    blocks, rows, d = k_ctx.shape
    assert d == 128 and rows % 8 == 0
    chunks = rows // 8
    means = torch.empty(blocks, chunks, 128, dtype=torch.bfloat16)
    mc = torch.empty(blocks, chunks, dtype=torch.bfloat16)
    lib().oracle_chunk_stats(_p(k_ctx), _l(rows * 128), _i(blocks), _i(chunks), _p(means), _p(mc))
    return means, mc
