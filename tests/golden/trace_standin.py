"""Recording stand-in for the native module `kernels.shadowkv` (TEST INFRASTRUCTURE).

The twelve names of the reference's pybind11 extension (/root/reference/kernels/main.cu:42-81) as Python functions
that (1) record every argument they are handed - scalars verbatim, tensors as (dtype, shape, stride, digest of the
bytes before the call, digest after it) - and (2) carry the call out on CPU tensors through the build's C
restatement (oracle/).  Two users:

  * tests/golden/make_golden.py registers it as `kernels.shadowkv` BEFORE loading the reference's own
    models/kv_cache.py + models/tensor_op.py, so the reference's `ShadowKVCache_CPU` decode methods
    (kv_cache.py:983-1176, 1227-1271) run offline, unmodified, and what they pass across boundary B2 is captured
    ("the strongest boundary test available offline");
  * tests/test_decode_trace.py patches it over `shadowkv_amd.kernels.shadowkv`'s functions and drives
    shadowkv_amd.kv_cache.ShadowKVCache_CPU (reference_calls=True) on the same seeded inputs: the two recordings
    must be equal call by call.

Nothing under shadowkv_amd/ imports this file.
"""
import hashlib
import types

import numpy as np
import torch

NAMES = ("gather_copy", "gather_copy_d2d_with_offsets", "reorder_keys_and_compute_offsets", "gather_copy_with_offsets",
         "apply_rotary_pos_emb", "apply_rotary_pos_emb_new", "apply_rotary_pos_emb_new_v2",
         "apply_rotary_pos_emb_push_cache", "apply_rotary_pos_emb_push_cache_opt",
         "apply_rotary_pos_emb_push_cache_opt_glm", "batch_gather_gemm", "batch_gemm_softmax")
# exported by the reference but never called from its Python: a call would be a finding
NEVER_CALLED = ("gather_copy", "apply_rotary_pos_emb", "apply_rotary_pos_emb_new_v2", "apply_rotary_pos_emb_push_cache")


def digest(t):
    """128 bits of the SHA-256 of a tensor's bytes (logical order), as hex."""
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        t = t.view(torch.int16)
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()[:32]


def describe(t):
    return {"dtype": str(t.dtype).replace("torch.", ""), "shape": list(t.shape), "stride": list(t.stride())}


class KernelTrace:
    """calls: list of {"fn", "tag", "args": [...]} where a tensor argument is {"dtype", "shape", "stride", "in", "out"}
    and a scalar argument is {"int": v} / {"float": v}.  `tag` is whatever the driver set with mark() (step, layer)."""

    def __init__(self, backend=None):
        if backend is None:
            import oracle as backend
        self.backend = backend
        self.calls = []
        self.tag = None

    def mark(self, tag):
        self.tag = tag

    def _wrap(self, name):
        def fn(*args):
            rec = {"fn": name, "tag": self.tag, "args": []}
            tens = []
            for a in args:
                if torch.is_tensor(a):
                    d = describe(a)
                    d["in"] = digest(a)
                    rec["args"].append(d)
                    tens.append((a, d))
                elif isinstance(a, bool) or not isinstance(a, (int, float)):
                    raise TypeError(f"{name}: argument of type {type(a).__name__} (the pybind signature takes tensors, "
                                    "ints and floats)")
                elif isinstance(a, int):
                    rec["args"].append({"int": int(a)})
                else:
                    rec["args"].append({"float": float(np.float32(a))})       # (a C float on the other side)
            if name in NEVER_CALLED:
                raise AssertionError(f"kernels.shadowkv.{name} is exported but the reference's Python never calls it")
            getattr(self.backend, name)(*args)
            for a, d in tens:
                d["out"] = digest(a)
            self.calls.append(rec)
        fn.__name__ = name
        return fn

    def module(self, name="kernels.shadowkv"):
        m = types.ModuleType(name)
        for n in NAMES:
            setattr(m, n, self._wrap(n))
        return m


def compare_calls(ref_calls, got_calls):
    """First difference between two recordings as a string, or None."""
    for i, (r, g) in enumerate(zip(ref_calls, got_calls)):
        where = f"call {i} ({r['fn']}, step/layer {r['tag']})"
        if r["fn"] != g["fn"]:
            return f"{where}: got {g['fn']}"
        if list(r["tag"]) != list(g["tag"]):
            return f"{where}: issued at step/layer {g['tag']}"
        if len(r["args"]) != len(g["args"]):
            return f"{where}: {len(g['args'])} arguments instead of {len(r['args'])}"
        for j, (ra, ga) in enumerate(zip(r["args"], g["args"])):
            if ra != ga:
                return f"{where}: argument {j}: reference {ra} != {ga}"
    if len(ref_calls) != len(got_calls):
        return f"{len(got_calls)} calls instead of {len(ref_calls)}"
    return None
