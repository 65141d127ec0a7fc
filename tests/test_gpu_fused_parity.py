"""GPU parity for the fused decode-path launchers (C ABI part 2) against the CPU oracle."""
import ctypes
import math

import pytest
import torch

import oracle
import os

import numpy as np

from util import check_attention, standalone_pass_labels, ALPHA, assert_bits_equal, ulp_diff_bf16, make_selection_step, check_topk_against_reference, record_parity, REBUILD_FLIP_BOUND

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from shadowkv_amd import _lib
    return _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _oracle_select(q, lm, lm_idx, cached, blocks, G, N, S):
    T = (N + 255) // 256
    D = torch.zeros(blocks, G, N, dtype=torch.bfloat16); P = torch.zeros_like(D)
    nm = torch.zeros(blocks, T, G); sm = torch.zeros(blocks, T, G)
    oracle.batch_gemm_softmax(q, lm, D, nm, sm, P, blocks, G, N, 128, ALPHA)
    sel = oracle.group_max_topk(P, lm_idx, blocks, G, N, S)
    c = cached.clone(); off = torch.zeros(blocks, S, dtype=torch.int32); cnt = torch.zeros(blocks, dtype=torch.int32)
    oracle.reorder_keys_and_compute_offsets(c, sel, off, cnt, 1, blocks, S)
    return P, sel, c, off, cnt


def _device_select(q, lm, lm_idx, cached, blocks, G, N, S):
    L = _lib()
    ws = torch.empty(L.lib().skv_select_workspace_bytes(blocks, G, N), dtype=torch.uint8, device=DEV)
    P = torch.zeros(blocks, G, N, dtype=torch.bfloat16, device=DEV)
    sel = torch.zeros(blocks, S, dtype=torch.int64, device=DEV)
    c = cached.to(DEV); off = torch.zeros(blocks, S, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(blocks, dtype=torch.int32, device=DEV)
    qd, lmd, lid = q.to(DEV), lm.to(DEV), (lm_idx.to(DEV) if lm_idx is not None else None)
    rc = L.lib().skv_select_chunks(qd.data_ptr(), lmd.data_ptr(), L.ptr(lid), c.data_ptr(), off.data_ptr(),
                                   cnt.data_ptr(), ws.data_ptr(), P.data_ptr(), sel.data_ptr(), blocks, G, N, S,
                                   ALPHA, _stream())
    L.check(rc, "skv_select_chunks")
    torch.cuda.synchronize()
    return P.cpu(), sel.cpu(), c.cpu(), off.cpu(), cnt.cpu()


@pytest.mark.parametrize("blocks,G,N,S", [(8, 4, 15560, 256), (8, 4, 504, 32), (4, 8, 25544, 256), (2, 1, 1000, 128),
                                           (3, 4, 300, 300), (2, 4, 131056, 256)])   # last: 1M-token context, score row
def test_select_chunks(blocks, G, N, S):                                              # too large for LDS staging
    g = torch.Generator().manual_seed(N + S)
    q = (torch.randn(blocks, G, 128, generator=g) * 3).bfloat16()
    lm = torch.randn(blocks, N, 128, generator=g).bfloat16()
    # landmark slot -> chunk id: increasing with gaps (outlier chunks removed), as prefill builds it
    lm_idx = torch.stack([torch.sort(torch.randperm(N + 48, generator=g)[:N]).values for _ in range(blocks)]).to(torch.int64)
    cached = torch.stack([lm_idx[b][torch.randperm(N, generator=g)[:S]] for b in range(blocks)])
    r0 = _oracle_select(q, lm, lm_idx, cached, blocks, G, N, S)
    r1 = _device_select(q, lm, lm_idx, cached, blocks, G, N, S)
    assert_bits_equal(r0[0], r1[0], "softmax P")
    assert torch.equal(r0[1], r1[1]), "selected chunk ids (top-k) differ"
    assert torch.equal(r0[4], r1[4]), "cnts differ"
    assert torch.equal(r0[2], r1[2]), "reordered position ids differ"
    assert torch.equal(r0[3], r1[3]), "offsets differ"


def _inplace_expected(cached, sel):
    """Model of the in-place layout: ids selected again keep their slot; the misses (ascending id) take the slots of the
    evicted ids (ascending slot).  Integer arithmetic only - the selected set `sel` comes from the oracle."""
    blocks, S = cached.shape
    exp = cached.clone(); cnts = []; miss_ids = []; dst = []
    for b in range(blocks):
        chosen = set(sel[b].tolist())
        resident = cached[b].tolist()
        keep = [i for i, cid in enumerate(resident) if cid in chosen and cid >= 0 and resident.index(cid) == i]
        kept_ids = {resident[i] for i in keep}
        free = [i for i in range(S) if i not in set(keep)]
        misses = sorted(chosen - kept_ids)
        assert len(free) == len(misses)
        for slot, cid in zip(free, misses):
            exp[b, slot] = cid
        cnts.append(len(keep)); miss_ids.append(misses); dst.append(free)
    return exp, cnts, miss_ids, dst


@pytest.mark.parametrize("blocks,G,N,S,overlap", [(8, 4, 15560, 256, 0.67), (4, 4, 4000, 256, 0.0), (4, 4, 4000, 256, 1.0),
                                                   (2, 8, 2000, 32, 0.5), (2, 4, 5000, 1024, 0.3), (3, 2, 300, 300, 1.0)])
def test_select_chunks_inplace(blocks, G, N, S, overlap):
    """skv_select_chunks_inplace: the selected SET is the oracle's (bit-exact top-k), the slot assignment follows the
    in-place rule exactly; all-hit (nothing written), all-miss (incl. the -1 initial state) and S = 1024 covered."""
    L = _lib()
    g = torch.Generator().manual_seed(7 * N + S)
    q = (torch.randn(blocks, G, 128, generator=g) * 3).bfloat16()
    lm = torch.randn(blocks, N, 128, generator=g).bfloat16()
    lm_idx = torch.stack([torch.sort(torch.randperm(N + 48, generator=g)[:N]).values for _ in range(blocks)]).to(torch.int64)
    _, sel, _, _, _ = _oracle_select(q, lm, lm_idx, lm_idx[:, :S].contiguous(), blocks, G, N, S)
    cached = torch.empty(blocks, S, dtype=torch.int64)
    for b in range(blocks):
        chosen = sel[b][torch.randperm(S, generator=g)]
        n_keep = int(round(overlap * S))
        others = torch.tensor(sorted(set(lm_idx[b].tolist()) - set(sel[b].tolist())))
        filler = others[torch.randperm(len(others), generator=g)[:S - n_keep]] if n_keep < S else others[:0]
        if overlap == 0.0 and b == 0:
            filler = torch.full((S,), -1, dtype=torch.int64)              # the cache's initial state
        row = torch.cat([chosen[:n_keep], filler])
        cached[b] = row[torch.randperm(S, generator=g)]
    exp, cnts, miss_ids, dst = _inplace_expected(cached, sel)
    ws = torch.empty(L.lib().skv_select_workspace_bytes(blocks, G, N), dtype=torch.uint8, device=DEV)
    c = cached.to(DEV)
    mids = torch.full((blocks, S), -7, dtype=torch.int32, device=DEV); slots = torch.full((blocks, S), -7, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(blocks, dtype=torch.int32, device=DEV); sel_out = torch.zeros(blocks, S, dtype=torch.int64, device=DEV)
    qd, lmd, lid = q.to(DEV), lm.to(DEV), lm_idx.to(DEV)
    L.check(L.lib().skv_select_chunks_inplace(qd.data_ptr(), lmd.data_ptr(), lid.data_ptr(), c.data_ptr(), mids.data_ptr(),
                                              slots.data_ptr(), cnt.data_ptr(), ws.data_ptr(), 0, sel_out.data_ptr(),
                                              blocks, G, N, S, S, 0, ALPHA, _stream()), "select_chunks_inplace")
    torch.cuda.synchronize()
    assert torch.equal(sel_out.cpu(), sel)
    assert cnt.cpu().tolist() == cnts
    assert torch.equal(c.cpu(), exp)
    for b in range(blocks):
        k = cnts[b]
        assert mids[b, k:].cpu().tolist() == miss_ids[b]
        assert slots[b, k:].cpu().tolist() == dst[b]
        assert torch.all(mids[b, :k] == -7)                                       # no miss id written for hits
        assert slots[b, :k].cpu().tolist() == sorted(set(range(S)) - set(dst[b]))  # [0, cnt): the hits' slots, ascending


def _select_from_scores(score, lm_idx, cached, S, inplace=False, age=None):
    """skv_select_from_scores on the device: (selected ids, cached after, offsets, cnts, dst_slots[, slot ages]).
    cached [blocks, R]; R > S (a resident set larger than the selection) needs inplace and the slot ages."""
    L = _lib()
    blocks, N = score.shape
    stride = (N + 7) // 8 * 8
    sc = torch.full((blocks, stride), 0x7f7f, dtype=torch.int16).view(torch.bfloat16)   # padding: large garbage, must be ignored
    sc[:, :N] = score
    scd = sc.to(DEV)
    c = cached.to(DEV); off = torch.full((blocks, S), -7, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(blocks, dtype=torch.int32, device=DEV); sel = torch.zeros(blocks, S, dtype=torch.int64, device=DEV)
    dst = torch.full((blocks, S), -7, dtype=torch.int32, device=DEV) if inplace else None
    lid = lm_idx.to(DEV) if lm_idx is not None else None
    aged = age.to(DEV) if age is not None else None
    L.check(L.lib().skv_select_from_scores(scd.data_ptr(), stride, L.ptr(lid), c.data_ptr(), off.data_ptr(), L.ptr(dst),
                                           cnt.data_ptr(), sel.data_ptr(), blocks, N, S, cached.shape[1], L.ptr(aged),
                                           _stream()), "select_from_scores")
    torch.cuda.synchronize()
    if age is not None:
        return sel.cpu(), c.cpu(), off.cpu(), cnt.cpu(), dst.cpu(), aged.cpu()
    return sel.cpu(), c.cpu(), off.cpu(), cnt.cpu(), (dst.cpu() if inplace else None)


def _oracle_from_scores(score, lm_idx, cached, S):
    blocks, N = score.shape
    sel = oracle.group_max_topk(score.view(blocks, 1, N).contiguous(), lm_idx, blocks, 1, N, S)
    c = cached.clone(); off = torch.zeros(blocks, S, dtype=torch.int32); cnt = torch.zeros(blocks, dtype=torch.int32)
    oracle.reorder_keys_and_compute_offsets(c, sel, off, cnt, 1, blocks, S)
    return sel, c, off, cnt


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", ["llama_small", "llama_cpu_b1024", "glm_small", "select_122k"])
def test_topk_stage_pinned_to_reference_scores(case):
    """The device's selection stage alone (skv_select_from_scores = the top-k / diff kernel of the decode path) on the
    bf16 scores the REFERENCE's torch.topk ran on (fixtures made by tests/golden/make_golden.py from the imported
    reference, models/kv_cache.py:421-445): same set as the reference modulo slots tied at the k-th value, identical
    where the boundary is unique; and identical to the oracle, reordering included.  select_122k = headline size."""
    z = np.load(os.path.join(GOLD, f"{case}.npz"))
    lm_idx = torch.from_numpy(z["lm_idx"][0].astype(np.int64)).contiguous()
    kv, N = lm_idx.shape
    g = torch.Generator().manual_seed(5)
    for t in range(z["sel"].shape[0]):
        score = torch.from_numpy(z["chunk_attn"][t][0].astype(np.int16)).view(torch.bfloat16).contiguous()
        ref_ids = torch.from_numpy(z["sel"][t][0])
        S = ref_ids.shape[-1]
        # resident set: the reference's previous selection (step 0: a random one) -> realistic hit / miss mix
        cached = (torch.from_numpy(z["sel"][t - 1][0]).clone() if t > 0
                  else torch.stack([lm_idx[h][torch.randperm(N, generator=g)[:S]] for h in range(kv)]))
        sel, c, off, cnt, _ = _select_from_scores(score, lm_idx, cached, S)
        check_topk_against_reference(score, lm_idx, ref_ids, sel, f"{case} step {t}")
        o = _oracle_from_scores(score, lm_idx, cached, S)
        assert torch.equal(sel, o[0]) and torch.equal(c, o[1]) and torch.equal(off, o[2]) and torch.equal(cnt, o[3])
        # in-place layout: same set
        sel2, c2, _, cnt2, _ = _select_from_scores(score, lm_idx, cached, S, inplace=True)
        assert torch.equal(sel2, sel) and torch.equal(cnt2, cnt)
        assert [sorted(r) for r in c2.tolist()] == [sorted(r) for r in c.tolist()]


@pytest.mark.parametrize("blocks,N,S,kind", [
    (3, 5000, 256, "wide"),        # k-th value > 32 binades below the maximum: the histogram window must slide
    (2, 9000, 256, "wide2"),       # two slides
    (2, 4096, 256, "flat"),        # every score equal: the whole row ties at the k-th value
    (2, 777, 777, "all"),          # S == N, N not a multiple of 8
    (2, 15560, 256, "steps"),      # few distinct values, k-th value tied thousands of times
    (1, 40000, 512, "seg8"),       # 8 vectors per thread
    (1, 131056, 256, "seg16"),     # 16 vectors per thread (1M-token context)
    (2, 300, 32, "zeros"),         # fewer non-zero scores than S: zeros tie at the k-th value
    (4, 15560, 256, "runs"),       # the top scores in runs of consecutive slots (attention locality): one thread owns up to 16
    (4, 15560, 256, "runs_tied"),  # ... with the k-th value tied inside the runs
    (2, 30000, 256, "runs"),       # 4 vectors (32 keys) per thread
])
def test_select_from_scores_edge_cases(blocks, N, S, kind):
    g = torch.Generator().manual_seed(N + S)
    if kind in ("wide", "wide2"):
        # 100 scores near 1, the rest spread over 1e-12 .. 1e-30 (wide2: 1e-25 .. 1e-36): the 256th value is far below
        lo, hi = (-30.0, -12.0) if kind == "wide" else (-36.0, -25.0)
        x = 10.0 ** (torch.rand(blocks, N, generator=g) * (hi - lo) + lo)
        top = torch.stack([torch.randperm(N, generator=g)[:100] for _ in range(blocks)])
        x.scatter_(1, top, torch.rand(blocks, 100, generator=g) * 0.5 + 0.5)
    elif kind == "flat":
        x = torch.full((blocks, N), 1.0 / N)
    elif kind == "steps":
        x = torch.tensor([1e-3, 3e-4, 1e-4, 6e-5])[torch.randint(0, 4, (blocks, N), generator=g)]
        x[:, ::97] = 0.02
    elif kind in ("runs", "runs_tied"):
        x = torch.softmax(torch.randn(blocks, N, generator=g) * 1.0, dim=-1) * 1e-2
        for bb in range(blocks):
            left = S + 7                                   # a little more than S high scores: the k-th value falls inside a run
            while left > 0:
                n = int(torch.randint(3, 33, (1,), generator=g))
                st = int(torch.randint(0, N - n, (1,), generator=g))
                vals = torch.rand(n, generator=g) * 0.5 + 0.5
                if kind == "runs_tied":
                    vals = torch.full((n,), 0.75)
                x[bb, st:st + n] = vals
                left -= n
    elif kind == "zeros":
        x = torch.zeros(blocks, N)
        x[:, ::29] = torch.rand(blocks, len(range(0, N, 29)), generator=g)
    else:
        x = torch.softmax(torch.randn(blocks, N, generator=g) * 3, dim=-1)
    score = x.bfloat16()
    lm_idx = torch.stack([torch.sort(torch.randperm(N + 48, generator=g)[:N]).values for _ in range(blocks)]).to(torch.int64)
    cached = torch.stack([lm_idx[b][torch.randperm(N, generator=g)[:S]] for b in range(blocks)])
    o = _oracle_from_scores(score, lm_idx, cached, S)
    r = _select_from_scores(score, lm_idx, cached, S)
    assert torch.equal(r[0], o[0]), "selected ids"
    assert torch.equal(r[1], o[1]) and torch.equal(r[2], o[2]) and torch.equal(r[3], o[3])


def _resident_expected(cached, age, sel, S):
    """Model of the resident-set policy (include/shadowkv_hip.h, skv_select_chunks_inplace): one head, one step.
    cached [R] ids per slot (-1 empty), age [R], sel = the S selected ids.  Returns (cached', age', cnt, miss ids,
    slots of the misses, slots of the hits)."""
    R = len(cached)
    pos = {}
    for slot, cid in enumerate(cached):
        if cid >= 0 and cid not in pos:
            pos[cid] = slot                                              # duplicates: the lowest slot answers
    hit_slots = sorted(pos[c] for c in sel if c in pos)
    misses = sorted(c for c in sel if c not in pos)
    eff = [63 if cached[s] < 0 else min(max(age[s], 0), 62) for s in range(R)]
    cand = [s for s in range(R) if s not in set(hit_slots)]
    evicted = sorted(sorted(cand, key=lambda s: (-eff[s], s))[:len(misses)])
    new_c, new_a = list(cached), [0] * R
    for s in range(R):
        if s in set(hit_slots) or s in set(evicted):
            new_a[s] = 0
        else:
            new_a[s] = 63 if cached[s] < 0 else min(eff[s] + 1, 62)
    for slot, cid in zip(evicted, misses):
        new_c[slot] = cid
    return new_c, new_a, len(hit_slots), misses, evicted, hit_slots


@pytest.mark.parametrize("blocks,N,S,R", [(4, 6000, 256, 512), (2, 3000, 64, 200), (2, 9000, 256, 1024), (2, 2000, 128, 129)])
def test_select_resident_set_larger_than_the_selection(blocks, N, S, R):
    """R resident slots per head, S selected per step (in-place layout): over a run of steps whose selections overlap,
    the slot -> id map, the slot ages, the hit counts, the miss ids, the slots the misses take (least recently selected
    first, ties and empty slots -> lowest slot) and the slots of the hits equal the policy model exactly; the selected
    SET is what the R == S kernel selects."""
    g = torch.Generator().manual_seed(N + R)
    lm_idx = torch.stack([torch.sort(torch.randperm(N + 48, generator=g)[:N]).values for _ in range(blocks)]).to(torch.int64)
    cached = torch.full((blocks, R), -1, dtype=torch.int64)
    age = torch.zeros(blocks, R, dtype=torch.int32)
    # start like a prefill: S slots filled, the rest empty; block 1 starts with old, saturated and tied ages
    for b in range(blocks):
        cached[b, :S] = lm_idx[b][torch.randperm(N, generator=g)[:S]]
    if blocks > 1:
        cached[1] = lm_idx[1][torch.randperm(N, generator=g)[:R]]
        age[1] = torch.randint(55, 70, (R,), generator=g, dtype=torch.int32)     # beyond 62: clamped
    logit = torch.randn(blocks, N, generator=g) * 2
    for step in range(10):
        logit = logit + torch.randn(blocks, N, generator=g) * (0.6 if step % 4 else 2.5)     # every 4th step: a jump
        score = torch.softmax(logit, dim=-1).bfloat16()
        sel_ref = _select_from_scores(score, lm_idx, cached[:, :S].contiguous(), S, inplace=True)[0]
        sel, c_new, off, cnt, dst, age_new = _select_from_scores(score, lm_idx, cached, S, inplace=True, age=age)
        assert torch.equal(sel, sel_ref), "a larger resident set must not change what is selected"
        for b in range(blocks):
            ec, ea, ecnt, emiss, eslots, ehit = _resident_expected(cached[b].tolist(), age[b].tolist(), sel[b].tolist(), S)
            assert int(cnt[b]) == ecnt, (step, b)
            assert c_new[b].tolist() == ec, (step, b)
            assert age_new[b].tolist() == ea, (step, b)
            assert off[b, ecnt:].tolist() == emiss and dst[b, ecnt:].tolist() == eslots and dst[b, :ecnt].tolist() == ehit
        cached, age = c_new, age_new
    assert int((cached >= 0).sum()) > blocks * S          # the set grew past the selection


def test_select_chunks_ties():
    """Massive ties at the threshold: duplicated landmarks give identical bf16 scores; the contract
    (lowest landmark slot wins) must hold bit-exactly."""
    blocks, G, N, S = 4, 4, 4096, 256
    g = torch.Generator().manual_seed(77)
    base = torch.randn(blocks, 37, 128, generator=g).bfloat16()
    lm = base[:, torch.randint(0, 37, (N,), generator=g)]  # only 37 distinct rows -> huge tie groups
    q = (torch.randn(blocks, G, 128, generator=g) * 0.5).bfloat16()
    cached = torch.stack([torch.randperm(N, generator=g)[:S] for _ in range(blocks)]).to(torch.int64)
    r0 = _oracle_select(q, lm.contiguous(), None, cached, blocks, G, N, S)
    r1 = _device_select(q, lm.contiguous(), None, cached, blocks, G, N, S)
    assert torch.equal(r0[1], r1[1])
    assert torch.equal(r0[2], r1[2]) and torch.equal(r0[3], r1[3]) and torch.equal(r0[4], r1[4])


@pytest.mark.parametrize("glm", [False, True])
def test_rebuild_keys(glm):
    g = torch.Generator().manual_seed(21 + glm)
    bs, heads, L, R, S, C = 1, 8, 8192, 160, 256, 8
    U = (torch.randn(bs, L, R, generator=g) / math.sqrt(R)).bfloat16()
    SV = torch.randn(bs, heads, 128, R, generator=g).bfloat16()
    ids = torch.stack([torch.randperm(L // C, generator=g)[:S] for _ in range(bs * heads)]).view(bs, heads, S).to(torch.int64)
    cnts = torch.randint(0, S + 1, (bs * heads,), generator=g, dtype=torch.int32)
    cnts[0] = 0; cnts[1] = S; cnts[2] = 3
    width = 64 if glm else 128
    cs = torch.randn(L + 64, width, generator=g).clamp(-1, 1).bfloat16()
    start, rows = 448, 448 + S * C + 96
    cache = torch.randn(bs, heads, rows, 128, generator=g).bfloat16()
    # oracle: gather-GEMM (bf16 round trip) then RoPE-push
    tmp = torch.zeros(bs, heads, S * C, 128, dtype=torch.bfloat16)
    ids32 = ids.to(torch.int32)
    oracle.batch_gather_gemm(U, SV, None, None, ids32, tmp, bs, heads, L, 128, R, S * C, L, C, torch.zeros_like(cnts))
    c0 = cache.clone()
    ints = (bs, heads, S * C, 128, tmp.stride(0), tmp.stride(1), tmp.stride(2), 1, cs.stride(0), ids32.stride(0),
            ids32.stride(1), ids32.stride(2), c0.stride(0), c0.stride(1), c0.stride(2), start, start + S * C, 64, C)
    (oracle.apply_rotary_pos_emb_push_cache_opt_glm if glm else oracle.apply_rotary_pos_emb_push_cache_opt)(
        tmp, cs, ids32, c0, cnts, *ints)
    L_ = _lib()
    c1 = cache.to(DEV)
    Ud, SVd, csd, idd, cnd = U.to(DEV), SV.to(DEV), cs.to(DEV), ids.to(DEV), cnts.to(DEV)
    rc = L_.lib().skv_rebuild_keys(Ud.data_ptr(), SVd.data_ptr(), csd.data_ptr(), idd.data_ptr(), cnd.data_ptr(),
                                   c1.data_ptr(), bs, heads, L, 128, R, S, C, cs.stride(0), c1.stride(0),
                                   c1.stride(1), c1.stride(2), start, 2 if glm else 1, 0, 0, _stream())
    L_.check(rc, "skv_rebuild_keys")
    torch.cuda.synchronize()
    c1 = c1.cpu()
    # everything outside the rebuilt rows must be untouched, bit for bit
    for b in range(bs):
        for h in range(heads):
            r0 = start + int(cnts[b * heads + h]) * C
            assert_bits_equal(c0[b, h, :r0], c1[b, h, :r0], "rows below the rebuilt range")
            assert_bits_equal(c0[b, h, start + S * C:], c1[b, h, start + S * C:], "rows above the sparse region")
    d = ulp_diff_bf16(c0, c1)
    # MFMA accumulates the 160 products in its own order, the oracle as a sequential chain, so a
    # small fraction of the pre-RoPE bf16 roundings flip by one ulp.  One flipped ulp of x1 or x2
    # (2^-8 relative) moves a rotated output by at most 2^-8*(|x1|+|x2|) plus its own roundings:
    # bound |diff| <= 2^-6 * (|x1| + |x2|) per rotation pair, and the flips must stay rare.
    frac, _ = record_parity(f"test_rebuild_keys[glm={glm}]", d[:, :, start:start + S * C], "post-RoPE")
    assert frac < REBUILD_FLIP_BOUND, f"{frac:.4f} of key values differ"
    x = tmp.float().abs()
    if glm:
        pair = x[..., 0:64:2] + x[..., 1:64:2]
        mag = torch.cat((torch.stack((pair, pair), -1).flatten(-2), x[..., 64:]), -1)
    else:
        pair = x[..., :64] + x[..., 64:]
        mag = torch.cat((pair, pair), -1)
    diff = (c0.float() - c1.float()).abs()[:, :, start:start + S * C]
    assert bool((diff <= 2.0 ** -6 * mag + 1e-6).all()), f"max excess {float((diff - 2.0 ** -6 * mag).max())}"


@pytest.mark.parametrize("bs,Hq,Hkv,kv_len,splits", [(1, 32, 8, 2497, 32), (2, 32, 8, 300, 8), (1, 32, 4, 2560, 32),
                                                     (1, 8, 8, 77, 16), (1, 32, 8, 5, 32),
                                                     # G = 8: the Q.K^T-on-MFMA body; ragged ranges, more splits than keys
                                                     (1, 16, 2, 77, 16), (2, 16, 2, 1000, 60), (1, 32, 4, 9, 32),
                                                     # more than 64 (batch, head) pairs: G = 4 takes Q.K^T on the MFMA with P.V
                                                     # on the VALU (HBM-bound batches), G = 8 stays all-MFMA
                                                     (9, 32, 8, 300, 3), (17, 32, 4, 131, 2)])
def test_sparse_attention(bs, Hq, Hkv, kv_len, splits):
    g = torch.Generator().manual_seed(kv_len)
    rows = kv_len + 50
    q = torch.randn(bs, Hq, 128, generator=g).bfloat16()
    k = torch.randn(bs, Hkv, rows, 128, generator=g).bfloat16()
    v = torch.randn(bs, Hkv, rows, 128, generator=g).bfloat16()
    scale = 1.0 / math.sqrt(128)
    labels = standalone_pass_labels(bs, Hq, Hkv, kv_len, splits)         # where this launch rounds its weights to bf16
    L = _lib()
    ws = torch.empty(L.lib().skv_attn_workspace_bytes(bs, Hq, splits), dtype=torch.uint8, device=DEV)
    out = torch.zeros(bs, Hq, 128, dtype=torch.bfloat16, device=DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    kv_dev = torch.tensor([kv_len], dtype=torch.int32, device=DEV)
    for kvp, host_len in ((0, kv_len), (kv_dev.data_ptr(), 0)):
        out.zero_()
        rc = L.lib().skv_sparse_attention(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                          kvp, host_len, rows, rows * 128, bs, Hq, Hkv, 128, splits, scale, _stream())
        L.check(rc, "skv_sparse_attention")
        torch.cuda.synchronize()
        # tolerance: fp16-level 1e-3 relative (north_star) + half a bf16 ulp of the output rounding, against the oracle that
        # rounds the softmax weights to bf16 where the all-MFMA pass does (small (batch, head) counts with G = 4 / 8 take it)
        check_attention(f"test_sparse_attention[{bs}-{Hq}-{Hkv}-{kv_len}-{splits}]", out.cpu().float(), q, k, v, kv_len, scale, labels)
    # kv_len past the rows a head owns: refused from the host, clamped from the device (never reads the next head)
    a = (qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), ws.data_ptr())
    assert L.lib().skv_sparse_attention(*a, 0, rows + 1, rows, rows * 128, bs, Hq, Hkv, 128, splits, scale, _stream()) == -1
    kv_dev.fill_(rows + 1000)
    L.check(L.lib().skv_sparse_attention(*a, kv_dev.data_ptr(), 0, rows, rows * 128, bs, Hq, Hkv, 128, splits, scale,
                                         _stream()), "skv_sparse_attention")
    torch.cuda.synchronize()
    check_attention(f"test_sparse_attention[{bs}-{Hq}-{Hkv}-{kv_len}-{splits}] clamped", out.cpu().float(), q, k, v, rows, scale,
                    standalone_pass_labels(bs, Hq, Hkv, rows, splits))


# ----------------------------------------------------------------------------------------------------------------------
# fused selection (round 4): scan -> top-k with the logit-domain prefilter, no normalise launch
# ----------------------------------------------------------------------------------------------------------------------
def _fused_case_inputs(blocks, G, N, kind, g):
    lm = torch.randn(blocks, N, 128, generator=g).bfloat16()
    if kind == "ties":
        lm[:, N // 8: N // 8 + 600] = lm[:, N // 8: N // 8 + 1]       # 600 identical landmarks: identical logits, tied bf16 scores
    if kind == "flat":
        lm[:] = lm[:, :1]                                             # every slot ties: the selection is the tie rule alone
    if kind == "runs":                                                # attention locality: the high scores sit in runs of neighbours
        base = torch.randn(blocks, 1, 128, generator=g)
        for b in range(blocks):
            for _ in range(12):
                st = int(torch.randint(0, N - 40, (1,), generator=g))
                lm[b, st:st + 32] = (base[b] * 2.0 + 0.3 * torch.randn(32, 128, generator=g)).bfloat16()
    lm_idx = torch.stack([torch.sort(torch.randperm(N + 48, generator=g)[:N]).values for _ in range(blocks)]).to(torch.int64)
    return lm, lm_idx


@pytest.mark.parametrize("blocks,G,N,S,kind,inplace", [
    (8, 4, 15560, 256, "walk", True),        # headline shape (2 key vectors per thread)
    (8, 4, 15560, 256, "walk", False),       # ... in the reference's slot order
    (4, 8, 25544, 256, "walk", True),        # GLM-4 / Yi-9B: G = 8, 4 key vectors per thread
    (2, 4, 31128, 512, "walk", True),        # budget 4096 at 244K
    (2, 4, 7500, 128, "walk", True),         # budget 1024 at 60K (1 key vector per thread)
    (3, 8, 777, 32, "walk", False),          # row shorter than the workgroup, N % 8 != 0
    (2, 4, 5000, 256, "ties", True),         # the k-th score tied hundreds of times: ties -> lowest slot
    (2, 8, 3000, 256, "flat", True),         # every slot a candidate: more than 2,048 -> every slot evaluated in the launch
    (4, 4, 15560, 256, "runs", True),        # a thread owns a run of candidates
    (2, 4, 9000, 256, "jump", True),         # a new query every step: the previous normalisers are useless
    (2, 4, 9000, 256, "garbage", False),     # the state holds garbage (+-1e3, inf): any finite or infinite value is valid input
])
def test_fused_selection_equals_the_three_launch_path(blocks, G, N, S, kind, inplace):
    """skv_select_chunks_fused against skv_select_chunks[_inplace] fed the same queries from the same resident set, step by
    step: selected ids, reordered / in-place resident ids, miss lists, destination slots and hit counts bit for bit - on the
    first step (normalisers 0: every slot is evaluated), on steps with good normalisers (fast path: S plus a few dozen
    candidates), on ties, on an all-equal row, after query jumps and with garbage in the state."""
    L = _lib()
    g = torch.Generator().manual_seed(11 * N + S + G)
    lm, lm_idx = _fused_case_inputs(blocks, G, N, kind, g)
    assert L.lib().skv_select_fused_supported(G, N, S) == 1
    lmd, lid = lm.to(DEV), lm_idx.to(DEV)
    ws_a = torch.empty(L.lib().skv_select_workspace_bytes(blocks, G, N), dtype=torch.uint8, device=DEV)
    ws_b = torch.empty_like(ws_a)
    state = torch.zeros(L.lib().skv_select_state_bytes(blocks, G), dtype=torch.uint8, device=DEV)
    L.check(L.lib().skv_select_state_init(state.data_ptr(), blocks, G, _stream()), "select_state_init")
    cached0 = torch.stack([lm_idx[b][torch.randperm(N, generator=g)[:S]] for b in range(blocks)])
    ca, cb = cached0.to(DEV), cached0.to(DEV)

    def bufs():
        return (torch.full((blocks, S), -7, dtype=torch.int32, device=DEV), torch.full((blocks, S), -7, dtype=torch.int32, device=DEV),
                torch.zeros(blocks, dtype=torch.int32, device=DEV), torch.zeros(blocks, S, dtype=torch.int64, device=DEV))
    q32 = torch.randn(blocks, G, 128, generator=g) * 2.0
    soff = L.lib().skv_select_state_stats_offset(blocks, G)
    paths = []
    for step in range(7):
        if kind == "jump" or step == 4:
            q32 = torch.randn(blocks, G, 128, generator=g) * (2.0 if step != 4 else 3.5)     # (step 4: a jump in every case)
        else:
            q32 = q32 + 0.3 * torch.randn(blocks, G, 128, generator=g)
        if kind == "garbage" and step in (2, 5):
            junk = torch.randn(blocks * G, generator=g) * 1e3
            junk[0] = float("inf"); junk[1] = float("-inf")
            state.view(torch.float32)[:blocks * G].copy_(junk)
            lvl0 = (blocks * G * 4 + 255) // 256 * 256 // 4          # the witness levels sit behind the normalisers (256-B aligned)
            state.view(torch.int32)[lvl0:lvl0 + blocks].copy_(torch.tensor([-5, 40000, 3, 2 ** 30][:blocks] + [0] * max(0, blocks - 4),
                                                                           dtype=torch.int32)[:blocks])
        qd = q32.bfloat16().to(DEV)
        ma, sa, na, oa = bufs()
        mb, sb, nb, ob = bufs()
        if inplace:
            L.check(L.lib().skv_select_chunks_inplace(qd.data_ptr(), lmd.data_ptr(), lid.data_ptr(), ca.data_ptr(), ma.data_ptr(),
                                                      sa.data_ptr(), na.data_ptr(), ws_a.data_ptr(), 0, oa.data_ptr(), blocks, G, N,
                                                      S, S, 0, ALPHA, _stream()), "select_chunks_inplace")
        else:
            L.check(L.lib().skv_select_chunks(qd.data_ptr(), lmd.data_ptr(), lid.data_ptr(), ca.data_ptr(), ma.data_ptr(),
                                              na.data_ptr(), ws_a.data_ptr(), 0, oa.data_ptr(), blocks, G, N, S, ALPHA, _stream()),
                    "select_chunks")
        L.check(L.lib().skv_select_chunks_fused(qd.data_ptr(), lmd.data_ptr(), lid.data_ptr(), cb.data_ptr(), mb.data_ptr(),
                                                sb.data_ptr() if inplace else 0, nb.data_ptr(), ws_b.data_ptr(), ob.data_ptr(),
                                                blocks, G, N, S, S, 0, ALPHA, state.data_ptr(), 0, 0, 0, 0, 0, 0.0, _stream()),
                "select_chunks_fused")
        torch.cuda.synchronize()
        assert torch.equal(oa, ob), f"{kind} step {step}: selected ids"
        assert torch.equal(na, nb), f"{kind} step {step}: hit counts"
        assert torch.equal(ca, cb), f"{kind} step {step}: resident ids"
        assert torch.equal(ma, mb), f"{kind} step {step}: offsets / miss ids"
        if inplace:
            assert torch.equal(sa, sb), f"{kind} step {step}: destination slots"
        st = state.view(torch.float32)[:blocks * G]
        assert bool(torch.isfinite(st).all()), "the state the launch leaves behind is this step's log-normalisers"
        stats = state[soff:soff + 8 * blocks].view(torch.int32).view(blocks, 2).cpu()
        paths.append(stats[:, 0].tolist())
        for path, ncand in stats.tolist():
            assert path in (0, 1, 2, 3) and (ncand > 2048) == bool(path & 2) or kind in ("ties", "flat")
            assert (path & 2) or S <= ncand <= 2048                # the candidates contain the selection
    # which branch ran (the results above are the same on all of them): no level on the first step -> searched
    assert all(p & 1 for p in paths[0])
    if kind == "flat":
        assert all(p & 2 for step in paths for p in step)          # N candidates > 2,048: every slot evaluated
    if kind in ("walk", "runs") and N >= 5000:
        held = sum(p == 0 for step in paths[1:4] + paths[5:] for p in step)
        assert held > 0, f"the carried level never held on a drifting query: {paths}"
    if kind == "flat":
        slot_of = [{int(c): j for j, c in enumerate(lm_idx[b].tolist())} for b in range(blocks)]
        for b in range(blocks):     # all scores equal: the tie rule alone decides - the S lowest slots
            assert sorted(slot_of[b][int(c)] for c in ob[b].tolist()) == list(range(S))


def test_fused_selection_refuses_unsupported_shapes():
    L = _lib().lib()
    assert L.skv_select_fused_supported(4, 15560, 256) == 1 and L.skv_select_fused_supported(8, 31128, 512) == 1
    assert L.skv_select_fused_supported(2, 15560, 256) == 0          # G not in {4, 8}: three-launch path
    assert L.skv_select_fused_supported(4, 40000, 256) == 0          # more than 32,768 landmarks per head
    assert L.skv_select_fused_supported(4, 100, 256) == 0
    assert L.skv_select_state_bytes(8, 4) >= 8 * 4 * 4 and L.skv_select_state_bytes(0, 4) == 0
