"""Drives a ShadowKVCache_CPU-shaped object (the reference's class in make_golden.py, shadowkv_amd's in the tests) through
prefill and TRACE_STEPS decode steps x layers in the order of LLM.layer_compute's decode branch
(/root/reference/models/base.py:315-341): update_kv_cache -> get_retrieval_position_ids -> get_value_cache ->
get_key_cache, and snapshots the observable state after every (step, layer).  TEST INFRASTRUCTURE."""
import math

import torch

import gen_inputs as G
from trace_standin import digest


def layer_inputs(case):
    c = G.TRACE_CASES[case]
    out = []
    for l in range(c["layers"]):
        per = []
        for b in range(c.get("batch", 1)):           # (sequence b of layer l: another draw; b = 0 is the single-sequence input)
            inp = G.make_inputs(case, layer=l + 100 * b)
            inp["k_roped"] = G.rope_torch(case, inp["k_pre"], inp["cos_sin"], torch.arange(c["L"]).unsqueeze(0))
            per.append(inp)
        inp = {k: (torch.cat([p[k] for p in per]) if k in ("k_pre", "v", "q_last", "k_roped") else per[0][k]) for k in per[0]}
        out.append(inp)
    return out


def prefill(cache, case, inputs, device="cpu"):
    """get_svd + prefill_kv_cache per layer as the prefill branch of layer_compute calls them (base.py:299-303), then H2D
    (base.py:620-627)."""
    for l, inp in enumerate(inputs):
        cache.get_svd(inp["k_pre"].to(device), l)
        cache.prefill_kv_cache(inp["v"].to(device), l, inp["k_roped"].to(device), inp["q_last"].to(device))
    cache.H2D()


def boundary_unique(lm, q, kv, groups, S):
    """bool [kv]: on that KV head the S-th largest group-max score is strictly above the (S+1)-th; scores as the decode
    path computes them (oracle.batch_gemm_softmax = the restated kernel, then the exact max over the group)."""
    import oracle
    N, D = lm.shape[-2], lm.shape[-1]
    T = (N + 255) // 256
    Dm = torch.zeros(kv, groups, N, dtype=torch.bfloat16)
    P = torch.zeros_like(Dm)
    oracle.batch_gemm_softmax(q.reshape(kv, groups, D).contiguous(), lm.reshape(kv, N, D).contiguous(), Dm,
                              torch.zeros(kv, groups, T), torch.zeros(kv, groups, T), P, kv, groups, N, D,
                              1 / math.sqrt(128), 0.0)
    score = P.float().max(dim=1).values
    srt = score.sort(dim=-1, descending=True).values
    return srt[:, S - 1] > srt[:, S]


def snapshot(cache, l, pos, v_view, k_view):
    return {"position_ids": cache.position_ids[l].cpu().flatten().tolist(),
            "offsets": cache.offsets.cpu().flatten().tolist(), "cnts": cache.cnts.cpu().flatten().tolist(),
            "signals": cache.signals.cpu().flatten().tolist(),
            "returned_ids": digest(pos), "k_buffer": digest(cache.k_cache_buffer[l]), "v_buffer": digest(cache.v_cache_buffer[l]),
            "k_view_shape": list(k_view.shape), "v_view_shape": list(v_view.shape),
            "k_view": digest(k_view), "v_view": digest(v_view),
            "kv_offset": int(cache.kv_offset), "gen_offset": int(cache.gen_offset), "kv_len": int(cache.get_kv_len())}


def decode(cache, case, inputs, trace=None, q_try=None, max_tries=4000, device="cpu", on_step=None):
    """q_try None: search the first tie-free draw per (step, layer, KV head) and return the numbers; else use the given ones.
    Returns (snapshots [step][layer], q_try [step][layer], q_digest [step][layer])."""
    c = G.TRACE_CASES[case]
    kv, groups, S = c["kv_heads"], c["q_heads"] // c["kv_heads"], c["budget"] // c["chunk"]
    bs = c.get("batch", 1)
    cos_sin = inputs[0]["cos_sin"].to(device)
    q_prev = [inp["q_last"] for inp in inputs]
    snaps, tries, qd = [], [], []
    for t in range(G.TRACE_STEPS):
        snaps.append([]); tries.append([]); qd.append([])
        for l in range(c["layers"]):
            if trace is not None:
                trace.mark([t, l])
            knew, vnew = G.trace_new_token(case, t, l)
            if q_try is None:
                lms = [cache.k_landmark[l][b].cpu() for b in range(bs)]
                a = [-1] * (bs * kv)
                for n in range(max_tries):
                    draw = G.trace_query_draw(case, q_prev[l], t, l, n)
                    for b in range(bs):
                        ok = boundary_unique(lms[b], draw[b:b + 1], kv, groups, S)
                        for h in range(kv):
                            if a[b * kv + h] < 0 and bool(ok[h]):
                                a[b * kv + h] = n
                    if min(a) >= 0:
                        break
                else:
                    raise RuntimeError(f"{case}: no tie-free query in {max_tries} draws at step {t} layer {l}: {a}")
            else:
                a = list(q_try[t][l])
            q = G.trace_query(case, q_prev[l], t, l, a)
            if q_try is None:
                assert all(bool(boundary_unique(lms[b], q[b:b + 1], kv, groups, S).all()) for b in range(bs))
            q_prev[l] = q
            tries[-1].append(a); qd[-1].append(digest(q))
            cache.update_kv_cache(knew.to(device), vnew.to(device), l)
            pos = cache.get_retrieval_position_ids(layer_idx=l, query_states=q.to(device))
            v_view = cache.get_value_cache(l, pos)
            k_view = cache.get_key_cache(layer_idx=l, position_ids=pos, rope_func=None, cos_sin_cache=cos_sin)
            snaps[-1].append(snapshot(cache, l, pos, v_view, k_view))
            if on_step is not None:
                on_step(t, l, q, pos, v_view, k_view)
    return snaps, tries, qd
