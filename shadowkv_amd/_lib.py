"""ctypes binding of libshadowkv_hip.so (C ABI: include/shadowkv_hip.h).

The product path has no CPU fallback: if the HIP library is missing this module raises at
first use, loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C shadowkv_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKV_LIB_PATH") or os.path.join(_HERE, "libshadowkv_hip.so")   # override: diagnostic builds only

_lib = None

c_int = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
c_p = ctypes.c_void_p
c_sz = ctypes.c_size_t

_ROPE_PUSH = [c_p] * 5 + [c_int] * 19 + [c_p]

_SIGS = {
    "skv_abi_version": (c_int, []),
    "skv_last_error": (ctypes.c_char_p, []),
    "skv_batch_gemm_softmax": (c_int, [c_p] * 6 + [c_int] * 4 + [c_f, c_f, c_p]),
    "skv_reorder_keys_and_compute_offsets": (c_int, [c_p] * 4 + [c_int] * 3 + [c_p]),
    "skv_gather_copy_with_offsets": (c_int, [c_p] * 6 + [c_int] * 7 + [c_p]),
    "skv_gather_copy_d2d_with_offsets": (c_int, [c_p] * 4 + [c_int] * 6 + [c_p]),
    "skv_gather_copy": (c_int, [c_p] * 3 + [c_int] * 5 + [c_p]),
    "skv_batch_gather_gemm": (c_int, [c_p] * 6 + [c_int] * 8 + [c_p, c_p]),
    "skv_apply_rotary_pos_emb_push_cache_opt": (c_int, _ROPE_PUSH),
    "skv_apply_rotary_pos_emb_push_cache_opt_glm": (c_int, _ROPE_PUSH),
    "skv_apply_rotary_pos_emb_push_cache": (c_int, _ROPE_PUSH),
    "skv_apply_rotary_pos_emb_new": (c_int, [c_p] * 4 + [c_int] * 13 + [c_p]),
    "skv_apply_rotary_pos_emb_new_v2": (c_int, [c_p] * 4 + [c_int] * 14 + [c_p]),
    "skv_apply_rotary_pos_emb": (c_int, [c_p] * 5 + [c_int] * 14 + [c_p]),
    "skv_select_workspace_bytes": (c_sz, [c_int] * 3),
    "skv_select_chunks": (c_int, [c_p] * 9 + [c_int] * 4 + [c_f, c_p]),
    "skv_fetch_kv_inplace_early": (c_int, [c_p] * 9 + [c_int] * 7 + [c_ll] * 4 + [c_int] * 2 + [c_ll] + [c_p] + [c_int] * 4 + [c_p]),
    "skv_select_chunks_early": (c_int, [c_p] * 9 + [c_int] * 4 + [c_f] + [c_p, c_p, c_ll, c_int, c_int, c_f, c_p]),
    "skv_fetch_kv_early": (c_int, [c_p] * 11 + [c_int] * 7 + [c_ll] * 4 + [c_int] * 2 + [c_ll] + [c_p] + [c_int] * 4 + [c_p]),
    "skv_select_chunks_inplace": (c_int, [c_p] * 10 + [c_int] * 5 + [c_p, c_f, c_p]),
    "skv_fetch_kv_inplace": (c_int, [c_p] * 9 + [c_int] * 7 + [c_ll] * 4 + [c_int] * 2 + [c_ll, c_p]),
    "skv_fetch_kv_attn_inplace": (c_int, [c_p] * 12 + [c_int] * 10 + [c_ll] * 4 + [c_int] * 2 + [c_ll, c_int, c_int, c_f, c_p]),
    "skv_select_fused_supported": (c_int, [c_int] * 3),
    "skv_select_state_bytes": (c_sz, [c_int] * 2),
    "skv_select_state_stats_offset": (c_sz, [c_int] * 2),
    "skv_select_state_init": (c_int, [c_p, c_int, c_int, c_p]),
    "skv_score_landmarks_fused": (c_int, [c_p] * 4 + [c_int] * 3 + [c_f, c_p, c_p, c_int, c_int, c_p]),
    "skv_select_chunks_fused": (c_int, [c_p] * 9 + [c_int] * 5 + [c_p, c_f] + [c_p, c_p, c_p, c_ll, c_int, c_int, c_f, c_p]),
    "skv_early_state_bytes": (c_sz, [c_int] * 5),
    "skv_early_state_offsets": (c_int, [c_int] * 5 + [ctypes.POINTER(c_ll)]),
    "skv_early_state_offsets2": (c_int, [c_int] * 5 + [ctypes.POINTER(c_ll), c_int]),
    "skv_early_state_init": (c_int, [c_p] + [c_int] * 5 + [c_p]),
    "skv_early_state_set_landmark_map": (c_int, [c_p, c_p] + [c_int] * 5 + [c_p]),
    "skv_select_chunks_inplace_early": (c_int, [c_p] * 10 + [c_int] * 5 + [c_p, c_f] + [c_p, c_p, c_ll, c_int, c_int, c_f] + [c_p]),
    "skv_fetch_kv_attn_inplace_early": (c_int, [c_p] * 12 + [c_int] * 10 + [c_ll] * 4 + [c_int] * 2 + [c_ll, c_int, c_int, c_f, c_p, c_int, c_int, c_int, c_p]),
    "skv_attn_finish_inplace": (c_int, [c_p] * 3 + [c_int] * 5 + [c_p]),
    "skv_sample_advance": (c_int, [c_p] * 2 + [c_int] * 2 + [c_f, ctypes.c_ulonglong] + [c_p] * 6 + [c_ll] * 3 + [c_p, c_int, c_p, c_p]),
    "skv_select_from_scores": (c_int, [c_p, c_int] + [c_p] * 6 + [c_int] * 4 + [c_p, c_p]),
    "skv_sample_topk_advance": (c_int, [c_p, c_ll] + [c_int] * 3 + [c_f, c_f, ctypes.c_ulonglong] + [c_p] * 6 + [c_ll] * 3 + [c_p, c_int, c_p, c_p]),
    "skv_sample_topk_advance_ranges": (c_int, [c_p, c_ll, c_int, c_p, c_ll] + [c_int] * 2 + [c_f, c_f, ctypes.c_ulonglong] + [c_p] * 6 + [c_ll] * 3 + [c_p, c_int, c_p, c_p]),
    "skv_score_landmarks": (c_int, [c_p] * 5 + [c_int] * 3 + [c_f, c_p]),
    "skv_score_landmarks_early": (c_int, [c_p] * 6 + [c_int] * 3 + [c_f, c_p, c_int, c_int, c_p]),
    "skv_rebuild_keys": (c_int, [c_p] * 6 + [c_int] * 7 + [c_ll] * 4 + [c_int] * 2 + [c_p] * 3),
    "skv_fetch_kv": (c_int, [c_p] * 11 + [c_int] * 7 + [c_ll] * 4 + [c_int] * 2 + [c_ll, c_p]),
    "skv_stage_hit_chunks": (c_int, [c_p] * 6 + [c_ll] * 2 + [c_int] * 2 + [c_p]),
    "skv_land_chunks": (c_int, [c_p] * 5 + [c_ll] * 3 + [c_int] * 2 + [c_p]),
    "skv_attn_workspace_bytes": (c_sz, [c_int] * 3),
    "skv_qkv_rope_update": (c_int, [c_p] * 8 + [c_int] * 4 + [c_ll, c_int] + [c_ll] * 2 + [c_int] * 2 + [c_p]),
    "skv_add_rmsnorm": (c_int, [c_p] * 5 + [c_int] * 2 + [c_f, c_p]),
    "skv_silu_and_mul": (c_int, [c_p] * 2 + [c_int] * 2 + [c_p]),
    "skv_update_kv_cache": (c_int, [c_p] * 4 + [c_int] * 4 + [c_ll] * 8 + [c_int] * 2 + [c_p]),
    "skv_norm_gemv_bf16": (c_int, [c_p] * 4 + [c_f] + [c_p] * 3 + [c_int] * 3 + [c_p]),
    "skv_norm_gemv_rangemax_bf16": (c_int, [c_p] * 4 + [c_f] + [c_p] * 3 + [c_int] * 2 + [c_p, c_p]),
    "skv_norm_gemv_near_pull_bf16": (c_int, [c_p] * 4 + [c_f] + [c_p] * 2 + [c_int] * 2 + [c_p] + [c_int] * 5 + [c_p, c_ll, c_int, c_int, c_p]),
    "skv_gemv_near_pull_bf16": (c_int, [c_p] * 4 + [c_int] * 2 + [c_p] + [c_int] * 5 + [c_p, c_ll, c_int, c_int, c_p]),
    "skv_qkv_gemv_rope_update": (c_int, [c_p] * 4 + [c_f] + [c_p] * 9 + [c_int] * 4 + [c_ll, c_int, c_ll] + [c_int] * 2 + [c_p]),
    "skv_gemv_bf16": (c_int, [c_p] * 4 + [c_int] * 3 + [c_p]),
    "skv_chunk_stats": (c_int, [c_p, c_ll] + [c_int] * 4 + [c_p] * 3),
    "skv_host_alloc": (c_int, [ctypes.POINTER(c_p), c_sz]),
    "skv_host_free": (c_int, [c_p]),
    "skv_linear_rows_bf16": (c_int, [c_p] * 4 + [c_int] * 4 + [c_p]),
    "skv_sparse_attention": (c_int, [c_p] * 6 + [c_int, c_int, c_ll] + [c_int] * 5 + [c_f, c_p]),
    "skv_sparse_attention_slots": (c_int, [c_p] * 6 + [c_int, c_int, c_ll] + [c_int] * 5 + [c_f, c_p] + [c_int] * 3 + [c_p]),
}

EXPORTS = tuple(_SIGS)


class ShadowKVNativeError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ShadowKVNativeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run __graft_entry__.build() or `make -C shadowkv_amd/csrc`). There is no CPU fallback.")
        # torch first: PyTorch-ROCm ships its own HIP runtime (libamdhip64 in torch/lib) and every stream / tensor this
        # library is handed comes from it.  Loaded the other way round, this library pulls in /opt/rocm's runtime first and
        # the process ends up with kernels registered in one runtime and streams from the other (observed: the first launch
        # that needs hipFuncSetAttribute fails).  With torch loaded, the NEEDED libamdhip64 resolves to the one already there.
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.skv_abi_version() != 1:
            raise ShadowKVNativeError("libshadowkv_hip.so ABI version mismatch")
        _lib = l
    return _lib


_ERR = {-1: "invalid argument (sizes / alignment)", -2: "shape not supported by the gfx950 kernels",
        -3: "kernel launch failed"}


def check(rc, what):
    if rc != 0:
        detail = lib().skv_last_error().decode() if rc == -3 else ""
        raise ShadowKVNativeError(f"{what}: {_ERR.get(rc, rc)} {detail}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = None


def current_stream_handle():
    """hipStream_t of torch's current stream on the current device (what `with torch.cuda.stream(...)` selects)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        # the raw getter skips the Stream object (called ~15 times per layer by the eager paths)
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
