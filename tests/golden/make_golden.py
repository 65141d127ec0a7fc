"""Fixture generator: drives the REFERENCE's own Python classes on seeded inputs and
stores what they produce under tests/golden/*.npz.

Runs only in the build container (needs /root/reference); the tests never import it and
never touch /root/reference.  No reference source is copied: the reference module is
loaded in place from /root/reference/models/kv_cache.py.

Loading recipe: `models/kv_cache.py` imports
`models.tensor_op` (pulls flashinfer/minference and allocates on "cuda" at import) and
`kernels.shadowkv` (the unbuilt CUDA extension).  Neither is used by the pure-PyTorch
class `ShadowKVCache`, nor by the prefill half of `ShadowKVCache_CPU`, so the module is
loaded alone with empty placeholders registered for those two names, and the few
torch.cuda calls it makes (Stream / synchronize / empty_cache / pin_memory=True) are
made no-ops on this CPU-only box.

What is captured per case (bf16 stored as uint16):
  pure-torch class `ShadowKVCache`   (models/kv_cache.py:155-506)
    svd_U, svd_SV            get_svd output                        (:278-317)
    lm, lm_idx               landmarks + their chunk ids           (:381-415)
    kbuf_head, vbuf_head     buffers [0, sparse_start)  (local + outlier rows)
    sel[t]                   selected chunk ids per decode step    (:421-445)
    vsparse[t], ksparse[t]   sparse region of V / K after step t   (:447-470)
    chunk_attn[t]            bf16 scores the top-k ran on
  offload class `ShadowKVCache_CPU`  (models/kv_cache.py:509-980), prefill half only
    cpu_*                    same state as above + initial position_ids and the
                             initial fill of the sparse region     (:921-970)
  offload class, DECODE half (round 5; trace_<case>.json, TRACE_CASES of gen_inputs.py)
    the reference's unmodified ShadowKVCache_CPU (models/kv_cache.py:983-1176, 1227-1271) and models/tensor_op.py:171-238
    over tests/golden/trace_standin.py registered as `kernels.shadowkv` (records every argument, carries the call out
    through oracle/): 2 layers x 4 steps in LLM.layer_compute's order - every call across the native boundary and the
    state after every (step, layer); trace_<case>_factors.npz: the reference's U / SV for two cases (LAPACK's bits
    differ between CPUs); run_trace / load_reference_offload below
  offload class, sub-batched prefill (subbatch_prefill.json): LLM.batch_prefill's pattern (models/base.py:533-543)

  python tests/golden/make_golden.py                 everything (~25 s)
  python tests/golden/make_golden.py --only-traces   the round-5 JSON fixtures only
Regeneration is deterministic: the JSON files come out byte-identical, the .npz arrays array-identical.
"""
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))        # repo root: `import oracle` (trace stand-in's back end)
import gen_inputs as G  # noqa: E402

REF = "/root/reference/models/kv_cache.py"


def load_reference_kv_cache():
    for name in ("models", "models.tensor_op", "kernels", "kernels.shadowkv"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["models.tensor_op"].batch_gather_gemm_rotary_pos_emb_cuda = None
    sys.modules["kernels"].shadowkv = sys.modules["kernels.shadowkv"]

    class _NoStream:
        def __init__(self, *a, **k):
            pass

    torch.cuda.Stream = _NoStream
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    _zeros = torch.zeros

    def zeros_nopin(*a, **k):
        k.pop("pin_memory", None)
        return _zeros(*a, **k)

    torch.zeros = zeros_nopin
    spec = importlib.util.spec_from_file_location("ref_kv_cache", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def u16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def run_case(ref, case):
    c = G.CASES[case]
    cfg = G.config_of(case)
    inp = G.make_inputs(case)
    L = c["L"]
    out = {}
    layer = 0

    # ---------------- pure-torch ShadowKVCache ----------------
    cache = ref.ShadowKVCache(cfg, batch_size=1, max_length=L, device="cpu", dtype=torch.bfloat16,
                              sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])
    cache.get_svd(inp["k_pre"], layer)
    pos_all = torch.arange(L).unsqueeze(0)
    k_roped = G.rope_torch(case, inp["k_pre"], inp["cos_sin"], pos_all)
    cache.prefill_kv_cache(inp["v"], layer, k_roped, inp["q_last"])
    out["svd_U"] = u16(cache.U[layer])
    out["svd_SV"] = u16(cache.SV[layer])          # [1, kv, rank, D]
    out["lm"] = u16(cache.k_landmark[layer])
    out["lm_idx"] = cache.k_landmark_idx[layer].numpy().copy()
    out["meta"] = np.array([cache.chunks, cache.prefill_local, cache.sparse_start, cache.sparse_end,
                            cache.select_sets, cache.outlier_chunk], dtype=np.int64)
    out["kbuf_head"] = u16(cache.k_cache_buffer[layer][:, :, :cache.sparse_start])
    out["vbuf_head"] = u16(cache.v_cache_buffer[layer][:, :, :cache.sparse_start])

    def rope_func(x, position_ids):
        return G.rope_torch(case, x, inp["cos_sin"], position_ids)

    sels, vs, ks, attns = [], [], [], []
    for t in range(inp["q_steps"].shape[0]):
        q = inp["q_steps"][t]
        # re-run the scoring by hand to capture the bf16 scores top-k ran on (same ops as :425-433)
        import math
        ca = torch.einsum('bhgqd,bhdc->bhgqc',
                          q.view(-1, cfg.num_key_value_heads, cache.num_key_value_groups, 1, 128),
                          cache.k_landmark[layer].transpose(2, 3)).squeeze(2) / math.sqrt(128)
        ca = torch.nn.functional.softmax(ca, dim=-1, dtype=torch.float32).to(torch.bfloat16).sum(dim=-2)
        if cache.num_key_value_groups > 1:
            ca, _ = torch.max(ca, dim=-2)
        attns.append(u16(ca))
        pos = cache.get_retrieval_position_ids(layer, q)
        sels.append(cache.selected_chunk_idx[layer].numpy().copy())
        v = cache.get_value_cache(layer, pos)
        k = cache.get_key_cache(layer, pos, rope_func, inp["cos_sin"])
        vs.append(u16(v[:, :, cache.sparse_start:cache.sparse_end]))
        ks.append(u16(k[:, :, cache.sparse_start:cache.sparse_end]))
    out["sel"] = np.stack(sels)
    out["vsparse"] = np.stack(vs)
    out["ksparse"] = np.stack(ks)
    out["chunk_attn"] = np.stack(attns)

    # ---------------- offload class, prefill half ----------------
    cpu = ref.ShadowKVCache_CPU(cfg, batch_size=1, max_length=L, device="cpu", dtype=torch.bfloat16,
                                sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])
    cpu.get_svd(inp["k_pre"], layer)
    cpu.prefill_kv_cache(inp["v"], layer, k_roped, inp["q_last"])
    out["cpu_SV"] = u16(cpu.SV[layer])            # [1, kv, D, rank]  (kernel layout, :730)
    out["cpu_U"] = u16(cpu.U[layer])
    out["cpu_lm"] = u16(cpu.k_landmark[layer])
    out["cpu_lm_idx"] = cpu.k_landmark_idx[layer].numpy().copy()
    out["cpu_meta"] = np.array([cpu.chunks, cpu.prefill_local, cpu.sparse_start, cpu.sparse_end,
                                cpu.select_sets, cpu.outlier_chunk, cpu.max_ctx_chunks_len,
                                cpu.kernel_offset, cpu.kernel_stride, cpu.kv_offset], dtype=np.int64)
    out["cpu_pos0"] = cpu.position_ids[layer].numpy().copy()
    out["cpu_kbuf"] = u16(cpu.k_cache_buffer[layer][:, :, :cpu.sparse_end])
    out["cpu_vbuf"] = u16(cpu.v_cache_buffer[layer][:, :, :cpu.sparse_end])
    # host V table: store only a checksum-like sample (full table is regenerable from v)
    nch = cpu.max_ctx_chunks_len // c["chunk"]
    out["cpu_vhost_rows"] = u16(cpu.v_cache_cpu[layer][:, :, [0, 1, nch // 2, nch - 1]])
    return out


# Arrays kept in full (inputs of later stages, or compared with a tolerance); everything
# else that is large is reduced to a SHA-256 digest `h_<name>` (byte-exact comparisons only).
KEEP_FULL = {
    "llama_small": {"svd_U", "svd_SV", "lm", "ksparse0", "cpu_lm"},
    "llama_cpu_b1024": {"lm"},
    "glm_small": {"svd_U", "svd_SV", "lm", "ksparse0"},
}
SMALL = 64 * 1024


def digest(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8).copy()


def shrink(case, out):
    out["ksparse0"] = out["ksparse"][0]
    res = {}
    for k, v in out.items():
        res["h_" + k] = digest(v)
        if k in KEEP_FULL[case] or v.nbytes <= SMALL:
            res[k] = v
    return res


def run_select_122k(ref):
    """Top-k stage at the headline size (BASELINE config 1: L = 124,928, budget 2,048 -> N = 15,560 landmarks, S = 256):
    the reference's pure-torch ShadowKVCache is prefilled on seeded synthetic K / V (inputs from gen_inputs, not stored)
    and its own get_retrieval_position_ids (models/kv_cache.py:421-445) runs for two queries.  Stored: the bf16 scores
    its torch.topk ran on (recomputed with the same ops, :425-433), the chunk ids it selected, and k_landmark_idx -
    i.e. inputs and outputs of the selection stage alone (torch.topk + gather, the stage kv_cache.py:1031-1042 repeats
    on the CUDA path)."""
    import math
    c = G.SELECT_122K
    cfg = SimpleNamespace(num_hidden_layers=1, num_attention_heads=c["q_heads"], num_key_value_heads=c["kv_heads"],
                          hidden_size=c["q_heads"] * c["head_dim"])
    inp = G.make_select_122k_inputs()
    cache = ref.ShadowKVCache(cfg, batch_size=1, max_length=c["L"], device="cpu", dtype=torch.bfloat16,
                              sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])
    cache.prefill_kv_cache(inp["v"], 0, inp["k_roped"], inp["q_steps"][0])
    kv, Gq = c["kv_heads"], c["q_heads"] // c["kv_heads"]
    attns, sels = [], []
    for t in range(inp["q_steps"].shape[0]):
        q = inp["q_steps"][t]
        ca = torch.einsum('bhgqd,bhdc->bhgqc', q.view(-1, kv, Gq, 1, 128),
                          cache.k_landmark[0].transpose(2, 3)).squeeze(2) / math.sqrt(128)
        ca = torch.nn.functional.softmax(ca, dim=-1, dtype=torch.float32).to(torch.bfloat16).sum(dim=-2)
        ca, _ = torch.max(ca, dim=-2)
        attns.append(u16(ca))
        cache.get_retrieval_position_ids(0, q)
        sels.append(cache.selected_chunk_idx[0].numpy().copy())
    lm_idx = cache.k_landmark_idx[0].numpy()
    return {"chunk_attn": np.stack(attns), "sel": np.stack(sels), "lm_idx": lm_idx.astype(np.int32),
            "meta": np.array([cache.chunks, cache.select_sets, cache.outlier_chunk, lm_idx.shape[-1]], dtype=np.int64)}


def load_reference_offload(kernels_shadowkv):
    """The reference's models/kv_cache.py AND its models/tensor_op.py, loaded in place, with `kernels.shadowkv` = the given
    module (tests/golden/trace_standin.py: records + carries the calls out through oracle/).  tensor_op.py imports
    flashinfer / minference (absent; none of their names is used on the decode path of the cache: empty placeholders) and
    builds one mask on "cuda" at import time (tensor_op.py:55-57; device dropped for the duration of the import).  What runs
    afterwards - ShadowKVCache_CPU's methods and tensor_op.batch_gather_gemm_rotary_pos_emb_cuda /
    apply_rotary_pos_emb_cuda_push_cache (tensor_op.py:171-238) - is the reference's own code, unmodified."""
    load_reference_kv_cache()                      # (torch.cuda / pin_memory no-ops)
    saved = {n: sys.modules.get(n) for n in ("models", "models.tensor_op", "kernels", "kernels.shadowkv", "flashinfer",
                                             "flashinfer.norm", "minference")}
    for name in ("models", "kernels", "flashinfer", "flashinfer.norm", "minference"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["kernels.shadowkv"] = kernels_shadowkv
    sys.modules["kernels"].shadowkv = kernels_shadowkv
    sys.modules["flashinfer.norm"].rmsnorm = None
    for n in ("vertical_slash_sparse_attention", "block_sparse_attention", "streaming_forward"):
        setattr(sys.modules["minference"], n, None)
    _arange = torch.arange

    def arange_nodev(*a, **k):
        k.pop("device", None)
        return _arange(*a, **k)

    try:
        torch.arange = arange_nodev
        spec = importlib.util.spec_from_file_location("models.tensor_op", "/root/reference/models/tensor_op.py")
        top = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(top)
    finally:
        torch.arange = _arange
    sys.modules["models.tensor_op"] = top
    spec = importlib.util.spec_from_file_location("ref_kv_cache_offload", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.batch_gather_gemm_rotary_pos_emb_cuda is top.batch_gather_gemm_rotary_pos_emb_cuda
    assert mod.shadowkv is kernels_shadowkv and top.shadowkv is kernels_shadowkv
    for n, m in saved.items():
        if m is None:
            sys.modules.pop(n, None)
        else:
            sys.modules[n] = m
    return mod


TRACE_FACTOR_CASES = ("trace_llama_b1024", "trace_glm_small")      # (trace_llama_b2048's U alone is 5.3 MB: not shipped)


def run_trace(case):
    """Decode half of the reference's ShadowKVCache_CPU (models/kv_cache.py:983-1176, 1227-1271): 2 layers x TRACE_STEPS
    steps in layer_compute's order, every call across `kernels.shadowkv` recorded (trace_standin.KernelTrace), state
    snapshots after every (step, layer)."""
    import trace_driver as TD
    from trace_standin import KernelTrace, digest as digest128
    c = G.TRACE_CASES[case]
    trace = KernelTrace()
    ref = load_reference_offload(trace.module())
    cache = ref.ShadowKVCache_CPU(G.config_of(case), batch_size=c.get("batch", 1), max_length=c["L"], device="cpu", dtype=torch.bfloat16,
                                  sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])
    inputs = TD.layer_inputs(case)
    TD.prefill(cache, case, inputs)
    assert not trace.calls, "prefill crossed the native boundary"
    meta = {"chunks": cache.chunks, "prefill_local": cache.prefill_local, "sparse_start": cache.sparse_start,
            "sparse_end": cache.sparse_end, "select_sets": cache.select_sets, "outlier_chunk": cache.outlier_chunk,
            "max_ctx_chunks_len": cache.max_ctx_chunks_len, "kernel_offset": cache.kernel_offset,
            "kernel_stride": cache.kernel_stride, "kv_offset": cache.kv_offset, "gen_offset": cache.gen_offset,
            "buffer_rows": cache.k_cache_buffer.shape[-2], "landmarks": cache.k_landmark.shape[-2]}
    state0 = {"U": digest128(cache.U), "SV": digest128(cache.SV), "k_landmark": digest128(cache.k_landmark),
              "k_landmark_idx": digest128(cache.k_landmark_idx), "position_ids": digest128(cache.position_ids),
              "k_cache_buffer": digest128(cache.k_cache_buffer), "v_cache_buffer": digest128(cache.v_cache_buffer),
              "v_cache_cpu": digest128(cache.v_cache_cpu)}
    if case in TRACE_FACTOR_CASES:
        # the reference's SVD factors themselves (bf16 as uint16): torch.svd goes through LAPACK, whose low-order bits depend on
        # the CPU model - a GPU box cannot regenerate them, and with them the recording's K bytes are reproducible anywhere
        np.savez_compressed(os.path.join(HERE, f"{case}_factors.npz"), U=u16(cache.U), SV=u16(cache.SV))
    snaps, tries, qd = TD.decode(cache, case, inputs, trace=trace)
    hits = [[sum(s["cnts"]) / (len(s["cnts"]) * cache.select_sets) for s in row] for row in snaps]
    return {"case": case, "meta": meta, "state_after_prefill": state0, "q_try": tries, "q_digest": qd,
            "chunk_hit_rate": hits, "snapshots": snaps, "calls": trace.calls}


STATE_NAMES = ("U", "SV", "k_landmark", "k_landmark_idx", "position_ids", "k_cache_buffer", "v_cache_buffer", "v_cache_cpu")


def run_subbatch(case):
    """The reference's ShadowKVCache_CPU prefilled the way LLM.batch_prefill does it (models/base.py:533-543): sub-batches of
    `sub` sequences, every layer per sub-batch (get_svd, prefill_kv_cache: kv_cache.py:683-737, 788-980)."""
    from trace_standin import digest as digest128
    c = G.SUBBATCH_CASES[case]
    ref = load_reference_kv_cache()
    cache = ref.ShadowKVCache_CPU(G.config_of(case), batch_size=c["batch"], max_length=c["L"], device="cpu", dtype=torch.bfloat16,
                                  sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])
    inputs = G.subbatch_inputs(case)
    progress = []
    for b0 in range(0, c["batch"], c["sub"]):
        sl = slice(b0, b0 + c["sub"])
        for l, inp in enumerate(inputs):
            cache.get_svd(inp["k_pre"][sl], l)
            cache.prefill_kv_cache(inp["v"][sl], l, inp["k_roped"][sl], inp["q_last"][sl])
        progress.append({"prefilled_batch": cache.prefilled_batch, "kv_offset": cache.kv_offset, "kv_len": cache.get_kv_len()})
    cache.H2D()
    return {"case": case, "progress": progress, "state": {n: digest128(getattr(cache, n)) for n in STATE_NAMES},
            "per_sequence": [{n: digest128(getattr(cache, n)[:, b:b + 1]) for n in STATE_NAMES} for b in range(c["batch"])],
            "meta": {"chunks": cache.chunks, "prefill_local": cache.prefill_local, "sparse_start": cache.sparse_start,
                     "sparse_end": cache.sparse_end, "landmarks": cache.k_landmark.shape[-2]}}


def write_subbatch(case):
    import json
    out = run_subbatch(case)
    path = os.path.join(HERE, "subbatch_prefill.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print(case, out["progress"], os.path.getsize(path), "B")


def write_trace(case):
    import json
    out = run_trace(case)
    path = os.path.join(HERE, f"{case}.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print(case, len(out["calls"]), "calls; q_try", out["q_try"], "hit rate",
          [[round(x, 2) for x in r] for r in out["chunk_hit_rate"]], os.path.getsize(path) // 1024, "KiB")


def main():
    if "--only-traces" in sys.argv:
        for case in G.TRACE_CASES:
            write_trace(case)
        for case in G.SUBBATCH_CASES:
            write_subbatch(case)
        return
    ref = load_reference_kv_cache()
    if "--only-small" not in sys.argv:
        out = run_select_122k(ref)
        path = os.path.join(HERE, "select_122k.npz")
        np.savez_compressed(path, **out)
        print("select_122k", {k: v.shape for k, v in out.items()}, os.path.getsize(path) // 1024, "KiB")
    for case in G.CASES:
        out = shrink(case, run_case(ref, case))
        path = os.path.join(HERE, f"{case}.npz")
        np.savez_compressed(path, **out)
        print(case, {k: v.shape for k, v in out.items()}, os.path.getsize(path) // 1024, "KiB")
    for case in G.TRACE_CASES:
        write_trace(case)
    for case in G.SUBBATCH_CASES:
        write_subbatch(case)


if __name__ == "__main__":
    main()
