"""GPU: the decode half of ShadowKVCache_CPU on MI355X against the recording of the REFERENCE's own decode half
(tests/golden/trace_*.json; how it was made and what the CPU side proves: tests/test_decode_trace.py).

The device cache starts from the prefill state the CPU build produces - byte-identical to the reference's (digests in
the fixture) - and is driven through 2 layers x 4 steps in LLM.layer_compute's order
(/root/reference/models/base.py:315-341).  Three forms of the same four methods, each against the fixture:
  * reference_calls=True: the reference's own launch sequence through the twelve `kernels.shadowkv` names on the HIP
    kernels (boundary B2 exactly as the reference crosses it);
  * the default methods (fused selection launch, staging + landing, fused rebuild) under the reference's copy_stream
    fork / join;
  * the default methods with lazy_value_fetch (+ the early fetch where the shape supports it).
Per (step, layer): position_ids / offsets / cnts / signals, the V buffer and V view (bit-exact, by digest), the views'
shapes, kv_offset / gen_offset equal the recording; the K buffer is compared with a CPU mirror that runs in lock-step
through the oracle (its digests ARE the recording's): rows outside the rebuilt range bit-exact, rebuilt rows within the
MFMA summation-order bound of tests/util.py (REBUILD_FLIP_BOUND, rope_pair_bound).
The recording's queries have a unique top-k boundary on every head, so any correct top-k selects the same set."""
import json
import os

import pytest
import torch

import gen_inputs as G
import trace_driver as TD
from trace_standin import KernelTrace, digest, NAMES
from util import assert_bits_equal, ulp_diff_bf16, record_parity, REBUILD_FLIP_BOUND

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STATE_TENSORS = ("U", "SV", "k_landmark", "k_landmark_idx")
STATE_SCALARS = ("prefill", "max_ctx_chunks_len", "chunks", "prefill_local", "sparse_start", "sparse_end", "kernel_offset",
                 "kernel_stride", "kv_offset", "gen_offset", "prefilled_batch")


def _new_cache(case, device):
    from shadowkv_amd.kv_cache import ShadowKVCache_CPU
    c = G.TRACE_CASES[case]
    return ShadowKVCache_CPU(G.config_of(case), batch_size=c.get("batch", 1), max_length=c["L"], device=device, dtype=torch.bfloat16,
                             sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])


def _transplant(src, dst):
    """The prefill state of `src` (built on the CPU: torch.svd through LAPACK, bit-pinned to the reference) into `dst`."""
    for n in STATE_TENSORS:
        setattr(dst, n, getattr(src, n).to(dst.device))
    for n in ("position_ids", "k_cache_buffer", "v_cache_buffer", "v_cache_cpu"):
        getattr(dst, n).copy_(getattr(src, n))
    for n in STATE_SCALARS:
        setattr(dst, n, getattr(src, n))
    dst.H2D()
    torch.cuda.synchronize()


class _Mirror:
    """CPU cache in reference_calls mode whose kernel calls go to the oracle (the stand-in), next to a device cache whose
    calls go to the HIP kernels: the twelve names dispatch on whether any tensor argument lives on the device."""

    def __init__(self, monkeypatch):
        import shadowkv_amd.kernels.shadowkv as K
        self.trace = KernelTrace()
        m = self.trace.module()
        for n in NAMES:
            real, standin = getattr(K, n), getattr(m, n)

            def f(*a, _r=real, _s=standin):
                on_device = any(torch.is_tensor(x) and x.is_cuda for x in a)      # (the V table is host memory on both sides)
                return (_r if on_device else _s)(*a)
            monkeypatch.setattr(K, n, f)


def _fixture(case):
    with open(os.path.join(GOLD, f"{case}.json")) as f:
        return json.load(f)


def _check_step_inplace(case, z, t, l, dev, mir):
    """The reference-shaped methods on the IN-PLACE layout (kv_cache.inplace_methods): hits keep their slots, so the slot order
    differs from the recording's - the chunk SET per head, the hit counts, the views' shapes and the bookkeeping must not; every
    slot holds the V rows of the chunk it names (== the mirror's rows of that chunk, which are the recording's)."""
    want = z["snapshots"][t][l]
    S, C = dev.select_sets, dev.chunk_size
    kv = dev.block_num                                              # (batch x KV heads)
    ids = dev.position_ids[l].flatten(0, 1).cpu()
    want_ids = torch.tensor(want["position_ids"]).view(kv, S)
    for h in range(kv):
        assert sorted(ids[h].tolist()) == sorted(want_ids[h].tolist()), f"{case} step {t} layer {l} head {h}: chunk set"
    assert dev.cnts.cpu().flatten().tolist() == want["cnts"], f"{case} step {t} layer {l}: cnts"
    for key, val in (("k_view_shape", list(dev._last_k_view.shape)), ("v_view_shape", list(dev._last_v_view.shape)),
                     ("kv_offset", int(dev.kv_offset)), ("gen_offset", int(dev.gen_offset)), ("kv_len", int(dev.get_kv_len()))):
        assert val == want[key], f"{case} step {t} layer {l}: {key}"
    assert digest(mir.v_cache_buffer[l]) == want["v_buffer"]
    vdev, vmir = dev.v_cache_buffer[l].flatten(0, 1).cpu(), mir.v_cache_buffer[l].flatten(0, 1)
    s0 = dev.sparse_start
    mir_ids = mir.position_ids[l].flatten(0, 1)
    for h in range(kv):
        slot_of = {int(c): j for j, c in enumerate(mir_ids[h].tolist())}
        for j, c in enumerate(ids[h].tolist()):
            a = vdev[h, s0 + j * C:s0 + (j + 1) * C]
            b = vmir[h, s0 + slot_of[c] * C:s0 + (slot_of[c] + 1) * C]
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"{case} step {t} layer {l} head {h} slot {j}: V rows of chunk {c}"
    assert_bits_equal(vmir[:, :s0], vdev[:, :s0], "local + outlier V rows")
    assert_bits_equal(vmir[:, dev.sparse_end:], vdev[:, dev.sparse_end:], "generated V rows")


def _check_step(case, z, t, l, dev, mir, k_diffs):
    if dev.inplace_methods:
        return _check_step_inplace(case, z, t, l, dev, mir)
    want = z["snapshots"][t][l]
    got = TD.snapshot(dev, l, dev.position_ids[l], dev._last_v_view, dev._last_k_view)
    for key in ("position_ids", "offsets", "cnts", "signals", "returned_ids", "v_buffer", "v_view", "k_view_shape",
                "v_view_shape", "kv_offset", "gen_offset", "kv_len"):
        assert got[key] == want[key], f"{case} step {t} layer {l}: {key}"
    # K: the mirror's buffer is the recording's (digest) when this box's LAPACK returns the recording's SVD factors (see
    # _drive); the device's differs from the mirror's by MFMA-order flips only
    kcpu = mir.k_cache_buffer[l].flatten(0, 1)                     # [batch x KV heads, rows, D]
    if mir.factors_pinned:
        assert digest(mir.k_cache_buffer[l]) == want["k_buffer"]
    kgpu = dev.k_cache_buffer[l].flatten(0, 1).cpu()
    s0, s1, C = dev.sparse_start, dev.sparse_end, dev.chunk_size
    assert_bits_equal(kcpu[:, :s0], kgpu[:, :s0], f"{case} step {t} layer {l}: local + outlier K rows")
    assert_bits_equal(kcpu[:, s1:], kgpu[:, s1:], f"{case} step {t} layer {l}: generated K rows")
    d = ulp_diff_bf16(kcpu[:, s0:s1], kgpu[:, s0:s1])
    frac, mx = record_parity(f"test_gpu_decode_trace[{case}] step {t} layer {l}", d, "post-RoPE")
    k_diffs.append(frac)
    # flips persist in hit rows (they keep the device's bits; the mirror keeps the oracle's): the bound covers the steps so far
    assert frac < REBUILD_FLIP_BOUND * (t + 1), f"{case} step {t} layer {l}: K sparse region differs in {frac} of the values"
    assert mx <= 2 or frac == 0.0 or bool(((kcpu[:, s0:s1].float() - kgpu[:, s0:s1].float()).abs()
                                           <= 2.0 ** -5 * kcpu[:, s0:s1].float().abs().amax() + 1e-6).all())


def _drive(case, monkeypatch, configure, streams):
    z = _fixture(case)
    _Mirror(monkeypatch)
    mir = _new_cache(case, "cpu")
    mir.reference_calls = True
    inputs = TD.layer_inputs(case)
    TD.prefill(mir, case, inputs)
    # Everything the prefill builds except the SVD factors is plain torch arithmetic and must equal the recording on any CPU;
    # torch.svd goes through LAPACK, whose kernels (and low-order bits) depend on the CPU model: on a box whose factors differ
    # from the recording's, K is still compared device-vs-oracle on THIS box's factors (same oracle code as the CPU test, which
    # pins oracle == reference on the recording's factors), V / ids / offsets / counts / bookkeeping against the recording.
    st0 = z["state_after_prefill"]
    for n in ("k_landmark", "k_landmark_idx", "position_ids", "k_cache_buffer", "v_cache_buffer", "v_cache_cpu"):
        assert digest(getattr(mir, n)) == st0[n], f"{case}: prefill state {n} differs from the recording"
    # -> for two of the three cases the recording's factors travel as a fixture (tests/golden/<case>_factors.npz) and replace
    # this CPU's: there the mirror's K bytes must equal the recording's at every step.
    fpath = os.path.join(GOLD, f"{case}_factors.npz")
    if os.path.exists(fpath):
        import numpy as np
        f = np.load(fpath)
        mir.U = torch.from_numpy(f["U"].astype(np.int16)).view(torch.bfloat16).clone()
        mir.SV = torch.from_numpy(f["SV"].astype(np.int16)).view(torch.bfloat16).clone()
        assert digest(mir.U) == st0["U"] and digest(mir.SV) == st0["SV"]
    mir.factors_pinned = digest(mir.U) == st0["U"] and digest(mir.SV) == st0["SV"]
    try:
        with open(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out", "decode_trace_factors.txt"), "a") as f:
            f.write(f"{case}: SVD factors of this CPU {'==' if mir.factors_pinned else '!='} the recording's\n")
    except OSError:
        pass
    dev = _new_cache(case, DEV)
    _transplant(mir, dev)
    configure(dev)
    c = G.TRACE_CASES[case]
    cos_dev = inputs[0]["cos_sin"].to(DEV)
    q_prev = [inp["q_last"] for inp in inputs]
    k_diffs = []
    for t in range(G.TRACE_STEPS):
        for l in range(c["layers"]):
            knew, vnew = G.trace_new_token(case, t, l)
            q = G.trace_query(case, q_prev[l], t, l, z["q_try"][t][l])
            q_prev[l] = q
            assert digest(q) == z["q_digest"][t][l]
            # mirror (CPU, oracle)
            mir.update_kv_cache(knew, vnew, l)
            pos = mir.get_retrieval_position_ids(layer_idx=l, query_states=q)
            mir.get_value_cache(l, pos)
            mir.get_key_cache(layer_idx=l, position_ids=pos, rope_func=None, cos_sin_cache=inputs[0]["cos_sin"])
            # device, in the reference's order (base.py:319-338)
            dev.update_kv_cache(knew.to(DEV), vnew.to(DEV), l)
            pos_d = dev.get_retrieval_position_ids(layer_idx=l, query_states=q.to(DEV))
            if streams:
                cur = torch.cuda.current_stream()
                with torch.cuda.stream(dev.copy_stream):
                    dev.copy_stream.wait_stream(cur)
                    dev._last_v_view = dev.get_value_cache(l, pos_d)
                dev._last_k_view = dev.get_key_cache(layer_idx=l, position_ids=pos_d, rope_func=None, cos_sin_cache=cos_dev)
                cur.wait_stream(dev.copy_stream)
            else:
                dev._last_v_view = dev.get_value_cache(l, pos_d)
                dev._last_k_view = dev.get_key_cache(layer_idx=l, position_ids=pos_d, rope_func=None, cos_sin_cache=cos_dev)
            torch.cuda.synchronize()
            _check_step(case, z, t, l, dev, mir, k_diffs)
    assert mir.factors_pinned or not os.path.exists(os.path.join(GOLD, f"{case}_factors.npz"))
    return dev, k_diffs


@pytest.mark.parametrize("case", list(G.TRACE_CASES))
def test_reference_launch_sequence_on_the_device(case, monkeypatch):
    def configure(dev):
        dev.reference_calls = True
    _drive(case, monkeypatch, configure, streams=True)


@pytest.mark.parametrize("case", list(G.TRACE_CASES))
def test_default_methods_on_the_device(case, monkeypatch):
    _drive(case, monkeypatch, lambda dev: None, streams=True)


@pytest.mark.parametrize("case", list(G.TRACE_CASES))
def test_deferred_value_fetch_and_early_fetch_on_the_device(case, monkeypatch):
    def configure(dev):
        dev.lazy_value_fetch = True
        if dev.early_fetch_supported():
            dev.enable_early_fetch()
    _drive(case, monkeypatch, configure, streams=True)


@pytest.mark.parametrize("case", list(G.TRACE_CASES))
def test_reference_shaped_methods_on_the_in_place_layout(case, monkeypatch):
    def configure(dev):
        dev.lazy_value_fetch = True
        dev.inplace_methods = True
    _drive(case, monkeypatch, configure, streams=True)
