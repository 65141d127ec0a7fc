"""Boundary B1/B2 pinned to the REFERENCE's own decode half.

tests/golden/trace_*.json were recorded by tests/golden/make_golden.py from the reference's unmodified
`ShadowKVCache_CPU` (/root/reference/models/kv_cache.py:983-1176, 1227-1271) and `models/tensor_op.py:171-238`, running
offline with `kernels.shadowkv` = tests/golden/trace_standin.py (records every argument, carries the call out through
oracle/): 2 layers x 4 decode steps in LLM.layer_compute's order, five cases (budget 1024; budget 2048 with the
headline's row layout: 48 outlier chunks, sparse region [448, 2496); BASELINE config 0: L = 4104, budget 256; GLM-4 shapes
with the width-64 cos/sin table; two sequences per cache).

CPU tests (this file, no GPU): shadowkv_amd.kv_cache.ShadowKVCache_CPU with reference_calls=True, the same stand-in patched
over shadowkv_amd.kernels.shadowkv, on the same seeded inputs ->
  * the recording (function order, every int / float, every tensor's dtype / shape / stride, the bytes of every tensor
    before and after the call) equals the reference's call by call: argument marshalling (cpu_v_length, gpu_v_offset /
    stride = kernel_offset / kernel_stride, the int32 cast of position_ids, the 19 ints of the RoPE push, ...);
  * the snapshots after every (step, layer) are equal: position_ids / offsets / cnts / signals, K and V buffers, the
    returned views' shapes ([: sparse_end + gen_offset (+ q_len unless last layer)], :1100,1172) and bytes,
    kv_offset / gen_offset advancing on the last layer only (:1269-1271).
The GPU counterpart (tests/test_gpu_decode_trace.py) runs the same drive on the device through the HIP kernels."""
import json
import os

import pytest
import torch

import gen_inputs as G
import trace_driver as TD
from trace_standin import KernelTrace, compare_calls, digest, NAMES

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(case):
    with open(os.path.join(GOLD, f"{case}.json")) as f:
        return json.load(f)


def patched_kernels(monkeypatch, trace):
    import shadowkv_amd.kernels.shadowkv as K
    m = trace.module()
    for n in NAMES:
        monkeypatch.setattr(K, n, getattr(m, n))


def build_cpu_cache(case):
    from shadowkv_amd.kv_cache import ShadowKVCache_CPU
    c = G.TRACE_CASES[case]
    cache = ShadowKVCache_CPU(G.config_of(case), batch_size=c.get("batch", 1), max_length=c["L"], device="cpu", dtype=torch.bfloat16,
                              sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])
    cache.reference_calls = True
    inputs = TD.layer_inputs(case)
    TD.prefill(cache, case, inputs)
    return cache, inputs


def state_digests(cache):
    return {"U": digest(cache.U), "SV": digest(cache.SV), "k_landmark": digest(cache.k_landmark),
            "k_landmark_idx": digest(cache.k_landmark_idx), "position_ids": digest(cache.position_ids),
            "k_cache_buffer": digest(cache.k_cache_buffer), "v_cache_buffer": digest(cache.v_cache_buffer),
            "v_cache_cpu": digest(cache.v_cache_cpu)}


@pytest.mark.parametrize("case", list(G.TRACE_CASES))
def test_decode_calls_and_state_equal_the_reference_recording(case, monkeypatch):
    z = load_fixture(case)
    cache, inputs = build_cpu_cache(case)
    meta = {"chunks": cache.chunks, "prefill_local": cache.prefill_local, "sparse_start": cache.sparse_start,
            "sparse_end": cache.sparse_end, "select_sets": cache.select_sets, "outlier_chunk": cache.outlier_chunk,
            "max_ctx_chunks_len": cache.max_ctx_chunks_len, "kernel_offset": cache.kernel_offset,
            "kernel_stride": cache.kernel_stride, "kv_offset": cache.kv_offset, "gen_offset": cache.gen_offset,
            "buffer_rows": cache.k_cache_buffer.shape[-2], "landmarks": cache.k_landmark.shape[-2]}
    assert meta == z["meta"]
    assert state_digests(cache) == z["state_after_prefill"]          # 2-layer prefill state == the reference's, byte for byte
    trace = KernelTrace()
    patched_kernels(monkeypatch, trace)
    snaps, tries, qd = TD.decode(cache, case, inputs, trace=trace, q_try=z["q_try"])
    assert qd == z["q_digest"]                                        # (the seeded queries are the ones the fixture was made with)
    diff = compare_calls(z["calls"], trace.calls)
    assert diff is None, diff
    for t, row in enumerate(z["snapshots"]):
        for l, want in enumerate(row):
            got = snaps[t][l]
            for key in want:
                assert got[key] == want[key], f"step {t} layer {l}: {key}"
    # what the recording itself says about the reference's bookkeeping (guards the fixture against a degenerate drive)
    last = z["snapshots"][-1][-1]
    assert last["gen_offset"] == G.TRACE_STEPS and last["kv_offset"] == z["meta"]["kv_offset"] + G.TRACE_STEPS
    assert z["snapshots"][0][0]["gen_offset"] == 0 and z["snapshots"][0][1]["gen_offset"] == 1     # last layer advances
    se = z["meta"]["sparse_end"]
    assert z["snapshots"][2][0]["k_view_shape"][2] == se + 2 + 1 and z["snapshots"][2][1]["k_view_shape"][2] == se + 3
    fns = [c["fn"] for c in z["calls"][:6]]
    assert fns[:5] == ["batch_gemm_softmax", "reorder_keys_and_compute_offsets", "gather_copy_with_offsets",
                       "gather_copy_d2d_with_offsets", "batch_gather_gemm"]
    assert fns[5] == ("apply_rotary_pos_emb_push_cache_opt_glm" if G.TRACE_CASES[case]["glm"]
                      else "apply_rotary_pos_emb_push_cache_opt")
    assert all(0.0 < x < 1.0 for row in z["chunk_hit_rate"] for x in row)      # every step mixes hits and misses


@pytest.mark.parametrize("case", ["trace_llama_b1024", "trace_glm_small"])
def test_shipped_factors_are_the_recordings(case):
    """tests/golden/<case>_factors.npz (the reference's U / SV, for GPU boxes whose LAPACK rounds differently): the digests the
    recording holds."""
    import numpy as np
    z = load_fixture(case)
    f = np.load(os.path.join(GOLD, f"{case}_factors.npz"))
    for n in ("U", "SV"):
        t = torch.from_numpy(f[n].astype(np.int16)).view(torch.bfloat16)
        assert digest(t) == z["state_after_prefill"][n]


def test_headline_row_layout_case_has_48_outliers():
    m = load_fixture("trace_llama_b2048")["meta"]
    assert (m["outlier_chunk"], m["prefill_local"], m["sparse_start"], m["sparse_end"], m["buffer_rows"],
            m["select_sets"]) == (48, 64, 448, 2496, 2592, 256)


def test_reference_calls_refuses_the_variants_the_reference_does_not_have():
    from shadowkv_amd.kv_cache import ShadowKVCache_CPU
    c = G.TRACE_CASES["trace_glm_small"]
    cache = ShadowKVCache_CPU(G.config_of("trace_glm_small"), batch_size=1, max_length=c["L"], device="cpu",
                              sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])
    cache.reference_calls = True
    cache.lazy_value_fetch = True
    with pytest.raises(RuntimeError):
        cache.get_value_cache(0, None)
