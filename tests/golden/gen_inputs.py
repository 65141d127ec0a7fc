"""Seeded synthetic inputs shared by make_golden.py (fixture generator, runs only where
/root/reference exists) and the tests (run anywhere).  Pure torch-CPU, no reference code.

The fixtures under tests/golden/*.npz hold only what cannot be regenerated from the
seed: tensors produced by the reference's own code (U/SV from its SVD, landmarks,
selections, buffers).  Inputs are regenerated here from the seed.
"""
import math
from types import SimpleNamespace

import torch

CASES = {
    # name: model-shape + cache-shape parameters
    "llama_small": dict(layers=1, q_heads=32, kv_heads=8, head_dim=128, L=2048, budget=256,
                        chunk=8, rank=160, rope_theta=500000.0, glm=False, seed=1234),
    "llama_cpu_b1024": dict(layers=1, q_heads=32, kv_heads=8, head_dim=128, L=2200, budget=1024,
                            chunk=8, rank=160, rope_theta=500000.0, glm=False, seed=4321),
    "glm_small": dict(layers=1, q_heads=32, kv_heads=4, head_dim=128, L=2048, budget=256,
                      chunk=8, rank=160, rope_theta=10000.0, glm=True, seed=777),
}


# ---- decode traces of the offload class (tests/golden/trace_*.json): 2 layers, the reference's ShadowKVCache_CPU driven
# through its four decode methods.  trace_llama_b2048 has the headline's row layout (48 outlier chunks, prefill_local 64,
# sparse region [448, 2496), 96 rows for generated tokens: SURVEY.md section 8) at a length torch.svd finishes in seconds.
TRACE_CASES = {
    "trace_llama_b1024": dict(layers=2, q_heads=32, kv_heads=8, head_dim=128, L=2200, budget=1024,
                              chunk=8, rank=160, rope_theta=500000.0, glm=False, seed=9101),
    "trace_llama_b2048": dict(layers=2, q_heads=32, kv_heads=8, head_dim=128, L=8256, budget=2048,
                              chunk=8, rank=160, rope_theta=500000.0, glm=False, seed=9202),
    "trace_glm_small": dict(layers=2, q_heads=32, kv_heads=4, head_dim=128, L=2048, budget=256,
                            chunk=8, rank=160, rope_theta=10000.0, glm=True, seed=9303),
    # BASELINE.json config 0 (the reference's own CPU-runnable case): 4K context (L > 4096: models/base.py:299 takes the prefill
    # branch), budget 256 -> 504 chunks, no outlier chunk, S = 32, 416 buffer rows
    "trace_llama_cfg0": dict(layers=2, q_heads=32, kv_heads=8, head_dim=128, L=4104, budget=256,
                             chunk=8, rank=160, rope_theta=500000.0, glm=False, seed=9606),
    # two sequences per cache: the batch dimension of every argument that crosses the native boundary (blocks = bs x kv heads)
    "trace_llama_bs2": dict(layers=2, q_heads=32, kv_heads=8, head_dim=128, L=2048, budget=256, batch=2,
                            chunk=8, rank=160, rope_theta=500000.0, glm=False, seed=9505),
}
TRACE_STEPS = 4


# ---- sub-batched prefill (tests/golden/subbatch_prefill.json): LLM.batch_prefill (models/base.py:500-548) prefills a batch
# in sub-batches of T sequences, every layer per sub-batch; the cache keeps `prefilled_batch` (kv_cache.py:683-737, 788-980).
SUBBATCH_CASES = {
    "subbatch_llama": dict(layers=2, q_heads=32, kv_heads=8, head_dim=128, L=2048, budget=256, chunk=8, rank=160,
                           rope_theta=500000.0, glm=False, seed=9404, batch=4, sub=2),
}


def case_of(case):
    for d in (CASES, TRACE_CASES, SUBBATCH_CASES):
        if case in d:
            return d[case]
    raise KeyError(case)


def subbatch_inputs(case):
    """Per layer: k_pre [B, L, kv*D] (the [bsz, seq, hidden] layout of get_svd, kv_cache.py:683), k_roped / v [B, kv, L, D],
    q_last [B, Hq, 1, D] - sequence b of layer l is make_inputs(case, layer=8 * l + b)."""
    c = SUBBATCH_CASES[case]
    out = []
    for l in range(c["layers"]):
        per = [make_inputs(case, layer=8 * l + b) for b in range(c["batch"])]
        pos = torch.arange(c["L"]).unsqueeze(0)
        k4 = torch.cat([p["k_pre"] for p in per])                                   # [B, kv, L, D]
        out.append(dict(k_pre=k4.transpose(1, 2).reshape(c["batch"], c["L"], -1).contiguous(),
                        k_roped=torch.cat([rope_torch(case, p["k_pre"], p["cos_sin"], pos) for p in per]),
                        v=torch.cat([p["v"] for p in per]), q_last=torch.cat([p["q_last"] for p in per]),
                        cos_sin=per[0]["cos_sin"]))
    return out


def config_of(case):
    c = case_of(case)
    return SimpleNamespace(num_hidden_layers=c["layers"], num_attention_heads=c["q_heads"],
                           num_key_value_heads=c["kv_heads"], hidden_size=c["q_heads"] * c["head_dim"])


def cos_sin_cache(case, max_pos):
    """Llama: [max_pos, 128] = cos[:64] | sin[:64] (models/llama.py:323-332 layout).
    GLM: [max_pos, 64] = cos[:32] | sin[:32] (models/glm.py:261-273 layout)."""
    c = case_of(case)
    rot = 64 if c["glm"] else c["head_dim"]
    inv_freq = 1.0 / (c["rope_theta"] ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    t = torch.arange(max_pos, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)  # [max_pos, rot/2]
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(torch.bfloat16).contiguous()


def make_inputs(case, layer=0):
    """Returns dict(k_pre [1,kv,L,D] bf16 pre-RoPE (approximately rank-`rank`), v [1,kv,L,D],
    q_last [1,q_heads,1,D], q_steps [steps,1,q_heads,1,D], cos_sin).  layer > 0: another draw (trace cases)."""
    c = case_of(case)
    g = torch.Generator().manual_seed(c["seed"] + 1000 * layer)
    L, kv, D, r = c["L"], c["kv_heads"], c["head_dim"], c["rank"]
    a = torch.randn(L, r, generator=g)
    b = torch.randn(r, kv * D, generator=g) / math.sqrt(r)
    k = a @ b + 0.05 * torch.randn(L, kv * D, generator=g)
    k_pre = k.view(1, L, kv, D).transpose(1, 2).contiguous().to(torch.bfloat16)
    v = torch.randn(1, kv, L, D, generator=g).to(torch.bfloat16)
    q_last = (torch.randn(1, c["q_heads"], 1, D, generator=g) * 2.0).to(torch.bfloat16)
    steps = 4
    q_steps = []
    qf = q_last.float()
    for _ in range(steps):
        qf = qf + 0.5 * torch.randn(qf.shape, generator=g)
        q_steps.append(qf.to(torch.bfloat16))
    return dict(k_pre=k_pre, v=v, q_last=q_last, q_steps=torch.stack(q_steps),
                cos_sin=cos_sin_cache(case, L + 1024))


def rope_neox_torch(x, cos_sin, position_ids):
    """bf16 tensor-op RoPE exactly as the reference's pure-torch helpers compute it
    (models/tensor_op.py:127-151): (x*cos) + (rotate_half(x)*sin), every op rounding to bf16.
    x [b,h,s,D]; position_ids [b,h,s] or [b,s]; cos_sin [P, D] = cos[:D/2] | sin[:D/2]."""
    half = cos_sin.shape[-1] // 2
    cos = torch.cat((cos_sin[:, :half], cos_sin[:, :half]), dim=-1)
    sin = torch.cat((cos_sin[:, half:], cos_sin[:, half:]), dim=-1)
    if position_ids.dim() == 2:
        position_ids = position_ids.unsqueeze(1).expand(-1, x.shape[1], -1)
    c = cos[position_ids]
    s = sin[position_ids]
    x1, x2 = x[..., :half], x[..., half:]
    rot = torch.cat((-x2, x1), dim=-1)
    return (x * c) + (rot * s)


def rope_glm_torch(x, cos_sin, position_ids):
    """GLM-4 interleaved half-dim RoPE in bf16 tensor ops: pairs (2t,2t+1), t<32 use
    cos_sin[pos, t], cos_sin[pos, t+32]; dims 64..127 pass through (models/glm.py:430-469)."""
    if position_ids.dim() == 2:
        position_ids = position_ids.unsqueeze(1).expand(-1, x.shape[1], -1)
    cs = cos_sin[position_ids]  # [b,h,s,64]
    c, s = cs[..., :32], cs[..., 32:]
    x_rot, x_pass = x[..., :64], x[..., 64:]
    xe, xo = x_rot[..., 0::2], x_rot[..., 1::2]
    oe = (xe * c) + ((-xo) * s)
    oo = (xo * c) + (xe * s)
    out_rot = torch.stack((oe, oo), dim=-1).flatten(-2)
    return torch.cat((out_rot, x_pass), dim=-1)


def rope_torch(case, x, cos_sin, position_ids):
    return rope_glm_torch(x, cos_sin, position_ids) if case_of(case)["glm"] else rope_neox_torch(x, cos_sin, position_ids)


def trace_new_token(case, step, layer):
    """The decoded token's K (as handed to update_kv_cache: post-RoPE, any bf16 values will do) and V, [bs, kv, 1, D]."""
    c = TRACE_CASES[case]
    ks, vs = [], []
    for b in range(c.get("batch", 1)):
        g = torch.Generator().manual_seed(c["seed"] * 31 + step * 17 + layer * 5 + 1 + 7919 * b)
        ks.append(torch.randn(1, c["kv_heads"], 1, c["head_dim"], generator=g).to(torch.bfloat16))
        vs.append(torch.randn(1, c["kv_heads"], 1, c["head_dim"], generator=g).to(torch.bfloat16))
    return torch.cat(ks), torch.cat(vs)


def trace_query_draw(case, q_prev, step, layer, attempt):
    """Random-walk query of (step, layer), draw number `attempt`: [bs, q_heads, 1, D] bf16 from the previous accepted one."""
    c = TRACE_CASES[case]
    g = torch.Generator().manual_seed(((c["seed"] * 131 + step) * 31 + layer) * 4099 + attempt)
    return (q_prev.float() + 0.5 * torch.randn(q_prev.shape, generator=g)).to(torch.bfloat16)


def trace_query(case, q_prev, step, layer, attempts):
    """The query of (step, layer): the q heads of KV group h of sequence b come from draw number attempts[b * kv + h].  The
    fixture generator takes, per (sequence, KV head), the first draw whose top-k boundary is unique (any correct top-k then
    returns the same SET - the reference leaves membership under ties undefined) and stores the numbers;
    the groups are independent."""
    c = TRACE_CASES[case]
    kv, groups, bs = c["kv_heads"], c["q_heads"] // c["kv_heads"], c.get("batch", 1)
    draws = {a: trace_query_draw(case, q_prev, step, layer, a) for a in set(attempts)}
    return torch.cat([torch.cat([draws[attempts[b * kv + h]][b:b + 1, h * groups:(h + 1) * groups] for h in range(kv)], dim=1)
                      for b in range(bs)], dim=0).contiguous()


# ---- selection stage at the headline size (tests/golden/select_122k.npz) ------------------------------------
SELECT_122K = dict(q_heads=32, kv_heads=8, head_dim=128, L=122 * 1024, budget=2048, chunk=8, rank=160, seed=2468)


def make_select_122k_inputs():
    """Post-RoPE keys with chunk structure (so landmarks carry signal and the scores have a realistic spread), V, and two
    queries aligned with a few chunks.  Only make_golden.py needs these (the fixture stores the scores themselves)."""
    c = SELECT_122K
    g = torch.Generator().manual_seed(c["seed"])
    L, kv, D, C = c["L"], c["kv_heads"], c["head_dim"], c["chunk"]
    centers = torch.randn(1, kv, L // C, 1, D, generator=g)
    k = (centers + 0.7 * torch.randn(1, kv, L // C, C, D, generator=g)).view(1, kv, L, D).to(torch.bfloat16)
    v = torch.randn(1, kv, L, D, generator=g).to(torch.bfloat16)
    qs = []
    for _ in range(2):
        q = torch.randn(1, c["q_heads"], 1, D, generator=g)
        pick = torch.randint(0, L // C, (c["q_heads"],), generator=g)
        for h in range(c["q_heads"]):
            q[0, h, 0] += 1.5 * centers[0, h // (c["q_heads"] // kv), pick[h], 0]
        qs.append((q * 1.2).to(torch.bfloat16))
    return dict(k_roped=k, v=v, q_steps=torch.stack(qs))
