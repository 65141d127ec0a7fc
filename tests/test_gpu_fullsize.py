"""GPU, BASELINE.json sizes, on the path the headline runs by default (bench.py: in-place layout, attention inside the fetch
launch, hipGraph, SPECULATIVE EARLY V FETCH ON with the default chunks per head): Llama-3.1-8B shapes at 124,928 tokens
(config 1: 61 scan tiles per head, E = 32) and GLM-4-9B shapes at 204,800 tokens (config 3: 4 KV heads x 8 query heads, GLM
RoPE, 100 scan tiles per head, E = 64), budget 2048, rank 160, 2 layers to keep the run short.
test_captured_default_path_against_the_oracle_at_full_size compares that path with the ORACLE step by step (selection set and
hit count per layer, V rows, attention); the other tests check invariants the domain offers over longer runs:
  * every slot of the sparse region holds exactly the V chunk its position_ids entry names (bytes from the host table:
    kv_cache.py:1081-1095 / copy.cuh:785-846 move byte-exact rows) - also for the chunks the early fetch staged in HBM
  * its K rows equal RoPE(U[rows] . SV^T) recomputed with PyTorch for the same ids (tolerance of the MFMA order)
  * position_ids of a head are distinct, in range, and none is an outlier chunk (all are landmark ids)
  * the K rebuild is homogeneous: scaling SV by 2 scales every rebuilt key by exactly 2 (bf16 scaling is exact and
    commutes with every rounding point), bit for bit, at full U size (64-bit row offsets)
  * with the early fetch on, tokens, position_ids, hit counts and both caches are bit-equal to the SAME captured steps
    without it, and chunks are pulled early on every layer in every step after the first
"""
import pytest
import torch

from util import rope_pair_bound

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# name: (model config, context tokens, sparse budget, landmarks per head, default early-fetch chunks per head)
SHAPES = {"llama31_122k": ("LLAMA_3_1_8B", 124928, 2048, 15560, 32), "glm4_200k": ("GLM_4_9B_1M", 204800, 2048, 25544, 64),
          # the reference's 244K regime (test/e2e.py:50-55): budget 4096 -> S = 512, 96 outlier chunks, 122 scan tiles per head,
          # 64 miss tiles + 24 splits = 88 attention records per head
          "llama31_244k_b4096": ("LLAMA_3_1_8B", 249856, 4096, 31128, 96)}
REPLAYS = 6
_RUNS = {}


def _run(shape, early, near=False):
    """One model of `shape` decoded for 2 eager warm-up steps + REPLAYS captured steps; cached per (shape, early, near) so that
    the comparison test reuses the fixtures' runs.  Returns (model, tokens of the replays, chunks pulled early per replay
    and layer or None)."""
    key = (shape, early, near)
    if key in _RUNS:
        return _RUNS[key]
    from shadowkv_amd import llama
    cfg_name, ctx, budget = SHAPES[shape][:3]
    m = llama.DecoderLM(cfg=getattr(llama, cfg_name), batch_size=1, max_length=ctx, device=DEV, sparse_budget=budget, rank=160,
                        chunk_size=8, num_layers=2, seed=3, chunk_layout="inplace", overlap_attention=True)
    llama.build_synthetic_context(m, ctx, seed=11)
    if early:
        m.kv_cache.enable_early_fetch(near=near)        # the default E of the shape: what bench.py's headline runs with
    table = llama.make_walk_table(m, 12, seed=5)
    dec = llama.GraphDecoder(m, temperature=0.6, walk_table=table)
    dec.token.copy_(torch.tensor([[7]], device=DEV))
    dec.capture()
    tokens, pulled = [], []
    m.near_served = 0
    for _ in range(REPLAYS):
        c = m.kv_cache
        ahead = [[set(r[r >= 0].tolist()) for r in c.near_published_ids(l)] for l in range(m.num_layers)] if near else None
        tokens.append(int(dec.step()[0, 0]))            # (reads the token back: one sync per step, as bench.py does)
        if early:
            pulled.append([int(m.kv_cache.early_fetch_counts(l).sum()) for l in range(m.num_layers)])
        if near:        # misses of the LAST layer (whose miss list the shared buffers still hold) served from chunks staged ahead
            l = m.num_layers - 1
            cnts, miss = c._cnts_layers[l].view(-1).cpu(), c.offsets.view(c.block_num, c.select_sets).cpu()
            m.near_served += sum(len(ahead[l][b] & set(miss[b, int(cnts[b]):].tolist())) for b in range(c.block_num))
    torch.cuda.synchronize()
    m.shape_name = shape
    _RUNS[key] = (m, tokens, pulled if early else None)
    return _RUNS[key]


@pytest.fixture(scope="module", params=[("llama31_122k", False), ("llama31_122k", True), ("glm4_200k", True), ("llama31_244k_b4096", True)],
                ids=["llama31_122k-plain", "llama31_122k-early32", "glm4_200k-early64", "llama31_244k_b4096-early96"])
def decoded(request):
    return _run(*request.param)[0]


def test_sparse_region_holds_the_chunks_its_ids_name(decoded):
    m = decoded
    c = m.kv_cache
    glm = m.cfg.rope_style == "glm"
    C, D, S = c.chunk_size, c.head_dim, c.select_sets
    budget = SHAPES[m.shape_name][2]
    assert (c.sparse_start, c.sparse_end, S) == {2048: (448, 2496, 256), 4096: (832, 4928, 512)}[budget]
    for l in range(m.num_layers):
        lm_ids = c.k_landmark_idx[l][0]                                   # [kv, N]
        for h in range(c.num_key_value_heads):
            ids = c.position_ids[l][0, h]
            assert ids.min() >= 0 and ids.max() < c.chunks and ids.unique().numel() == S
            assert torch.isin(ids, lm_ids[h]).all()                        # only landmark (non-outlier) chunks
            want_v = c.v_cache_cpu[l][0, h][ids.cpu()].to(DEV).view(S * C, D)
            got_v = c.v_cache_buffer[l][0, h, c.sparse_start:c.sparse_end]
            assert torch.equal(got_v.view(torch.int16), want_v.view(torch.int16)), (l, h)
            # K rows: PyTorch f32 reconstruction + RoPE of the same token rows
            tok = (ids.unsqueeze(-1) * C + torch.arange(C, device=DEV)).view(-1)
            k_pre = (c.U[l][0][tok].float() @ c.SV[l][0, h].float().t()).bfloat16()   # [S*C, D]
            cs = m.cos_sin_cache[tok].float()
            if glm:       # interleaved pairs (2t, 2t+1) of dims 0..63 with cos = cs[t], sin = cs[32 + t]; 64..127 copied
                cos, sin = cs[:, :32], cs[:, 32:]
                xe, xo = k_pre[:, 0:64:2].float(), k_pre[:, 1:64:2].float()
                rot = torch.stack((xe * cos - xo * sin, xo * cos + xe * sin), dim=-1).flatten(-2)
                want_k = torch.cat((rot, k_pre[:, 64:].float()), dim=-1)
            else:
                cos, sin = cs[:, :64], cs[:, 64:]
                x1, x2 = k_pre[:, :64].float(), k_pre[:, 64:].float()
                want_k = torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1)
            got_k = c.k_cache_buffer[l][0, h, c.sparse_start:c.sparse_end].float()
            bound = rope_pair_bound(k_pre, glm).to(DEV) + 1e-3
            assert bool(((got_k - want_k).abs() <= bound).all()), (l, h, float(((got_k - want_k).abs() - bound).max()))


def test_generated_rows_and_counters(decoded):
    m = decoded
    c = m.kv_cache
    n = 2 + REPLAYS                                                        # 2 eager warm-up steps + the replays
    ctx = SHAPES[m.shape_name][1]
    assert c.kv_offset == ctx + n and c.gen_offset == n
    for l in range(m.num_layers):
        for buf in (c.k_cache_buffer, c.v_cache_buffer):
            gen = buf[l][0, :, c.sparse_end:c.sparse_end + n].float()
            assert torch.isfinite(gen).all() and bool((gen.abs().sum(dim=-1) > 0).all())
            assert float(buf[l][0, :, c.sparse_end + n:].float().abs().sum()) == 0.0


def test_rebuild_is_homogeneous_at_full_size(decoded):
    from shadowkv_amd import tensor_op
    m = decoded
    c = m.kv_cache
    l = 0
    ids = c.position_ids[l].clone()
    ids[0, 0, :4] = torch.tensor([c.chunks - 1, c.chunks - 2, 0, 1], device=DEV)      # both ends of U (64-bit offsets)
    cnts = torch.zeros_like(c.cnts)
    out = []
    for scale in (1.0, 2.0):
        buf = torch.zeros_like(c.k_cache_buffer[l])
        tensor_op.rebuild_keys(c.U[l], (c.SV[l].float() * scale).bfloat16(), m.cos_sin_cache, ids, cnts, buf,
                               c.sparse_start, c.chunk_size)
        out.append(buf[:, :, c.sparse_start:c.sparse_end].float())
    torch.cuda.synchronize()
    assert float(out[0].abs().sum()) > 0
    assert torch.equal(out[1], 2.0 * out[0])


@pytest.mark.parametrize("shape", ["llama31_122k", "glm4_200k", "llama31_244k_b4096"])
def test_early_fetch_changes_no_bit_at_full_size(shape):
    """The headline's default path against the same captured steps without the early fetch, at the size the headline runs
    it (61 / 100 flag tiles per head, early_of over 15,616 / 25,600 chunks): sampled tokens, slot -> chunk map, hit counts
    and both caches bit for bit; the prediction fires on every layer in every step after the first (the two eager warm-up
    steps precede the replays, so every replay has thresholds)."""
    me, tok_e, pulled = _run(shape, True)
    mp, tok_p, _ = _run(shape, False)
    ce, cp = me.kv_cache, mp.kv_cache
    E = ce._early["E"]
    assert E == SHAPES[shape][4] and ce.k_landmark.shape[-2] == SHAPES[shape][3]
    assert tok_e == tok_p, (tok_e, tok_p)
    assert torch.equal(ce.position_ids, cp.position_ids)
    assert torch.equal(ce._cnts_layers, cp._cnts_layers)
    assert torch.equal(ce.v_cache_buffer.view(torch.int16), cp.v_cache_buffer.view(torch.int16))
    assert torch.equal(ce.k_cache_buffer.view(torch.int16), cp.k_cache_buffer.view(torch.int16))
    assert len(pulled) == REPLAYS
    for step, per_layer in enumerate(pulled):
        assert all(0 < n <= E * ce.block_num for n in per_layer), (step, per_layer)
    misses = ce.block_num * ce.select_sets * me.num_layers - int(ce._cnts_layers.sum())
    assert misses > 0, "the walk produced no miss in the last step: nothing was fetched"


@pytest.mark.parametrize("shape", ["llama31_122k", "glm4_200k"])
def test_near_miss_staging_changes_no_bit_at_full_size(shape):
    """Round 5: the captured step with the gate/up launches staging near misses ahead of the next step, against the same
    captured steps with the early fetch alone, at the headline sizes: sampled tokens, slot -> chunk map, hit counts and both
    caches bit for bit; and misses ARE served from the chunks staged ahead."""
    mn, tok_n, _ = _run(shape, True, near=True)
    me, tok_e, _ = _run(shape, True)
    cn, ce = mn.kv_cache, me.kv_cache
    assert cn.near_fetch and not ce.near_fetch
    assert tok_n == tok_e, (tok_n, tok_e)
    assert torch.equal(cn.position_ids, ce.position_ids)
    assert torch.equal(cn._cnts_layers, ce._cnts_layers)
    assert torch.equal(cn.v_cache_buffer.view(torch.int16), ce.v_cache_buffer.view(torch.int16))
    assert torch.equal(cn.k_cache_buffer.view(torch.int16), ce.k_cache_buffer.view(torch.int16))
    assert mn.near_served > 0
    for l in range(mn.num_layers):
        assert int((cn.near_published_ids(l) >= 0).sum()) > 0


@pytest.mark.parametrize("shape", ["llama31_122k", "glm4_200k"])
def test_captured_default_path_against_the_oracle_at_full_size(shape):
    """BASELINE.json configs 1 and 3 on the exact path bench.py times (captured step, fused selection, in-place layout, early
    fetch with the shape's default E, attention inside the fetch launch), 2 layers, 2 eager warm-up steps + 3 replays, against
    the oracle on the walk table's queries (about 30 ms of oracle per layer and step on the GPU box's cores):
      * per layer and step the resident chunk SET == oracle.batch_gemm_softmax + group_max_topk (bit-exact scores, the same
        tie rule), and the hit count == |selection & previous selection|;
      * every slot's V rows == the host table's rows of the chunk it names, byte for byte (all layers, every step);
      * the last layer's attention output == the oracle's over the device's own K / V bytes (check_attention with the
        rounding labels of the fetch launch's miss tiles; `_dst_slots` / `cnts` hold the last layer's bookkeeping)."""
    import math
    import oracle
    from shadowkv_amd import llama
    from util import check_attention, overlapped_pass_labels, ALPHA
    cfg_name, ctx, budget, n_lm, E = SHAPES[shape]
    m = llama.DecoderLM(cfg=getattr(llama, cfg_name), batch_size=1, max_length=ctx, device=DEV, sparse_budget=budget, rank=160,
                        chunk_size=8, num_layers=2, seed=3, chunk_layout="inplace", overlap_attention=True)
    llama.build_synthetic_context(m, ctx, seed=11)
    c = m.kv_cache
    c.enable_early_fetch()
    assert c._early["E"] == E and c.fused_select and c.k_landmark.shape[-2] == n_lm
    kv, Hq, D, S, C = c.num_key_value_heads, c.num_attention_heads, c.head_dim, c.select_sets, c.chunk_size
    G = Hq // kv
    c.attn_out_tap = [torch.zeros(1, 1, Hq, D, dtype=torch.bfloat16, device=DEV) for _ in range(m.num_layers)]
    table = llama.make_walk_table(m, 12, seed=5)                              # [T, L, 1, Hq, 1, D]
    lm = [c.k_landmark[l][0].cpu().contiguous() for l in range(m.num_layers)]
    lm_idx = [c.k_landmark_idx[l][0].cpu().contiguous() for l in range(m.num_layers)]
    T = (n_lm + 255) // 256
    resident = [[set(c.position_ids[l][0, h].tolist()) for h in range(kv)] for l in range(m.num_layers)]

    def oracle_step(i):
        """selection of step i per layer (sets per head) and the hit counts against the resident sets; advances them"""
        hits = []
        for l in range(m.num_layers):
            q = table[i % table.shape[0]][l][0].cpu().view(kv, G, D).contiguous()
            Dm = torch.zeros(kv, G, n_lm, dtype=torch.bfloat16); P = torch.zeros_like(Dm)
            oracle.batch_gemm_softmax(q, lm[l], Dm, torch.zeros(kv, T, G), torch.zeros(kv, T, G), P, kv, G, n_lm, D, ALPHA)
            sel = oracle.group_max_topk(P, lm_idx[l], kv, G, n_lm, S)
            new = [set(sel[h].tolist()) for h in range(kv)]
            assert all(len(x) == S for x in new)
            hits.append([len(new[h] & resident[l][h]) for h in range(kv)])
            resident[l] = new
        return hits

    def check_state(i, hits, what):
        for l in range(m.num_layers):
            ids = c.position_ids[l][0].cpu()
            for h in range(kv):
                assert set(ids[h].tolist()) == resident[l][h], f"{shape} {what} layer {l} head {h}: resident set != oracle selection"
            assert c._cnts_layers[l].cpu().tolist() == hits[l], f"{shape} {what} layer {l}: hit counts"
            want_v = torch.stack([c.v_cache_cpu[l][0, h][ids[h]] for h in range(kv)]).view(kv, S * C, D)
            got_v = c.v_cache_buffer[l][0, :, c.sparse_start:c.sparse_end].cpu()
            assert torch.equal(got_v.view(torch.int16), want_v.view(torch.int16)), f"{shape} {what} layer {l}: V rows"

    dec = llama.GraphDecoder(m, temperature=0.6, walk_table=table)
    dec.token.copy_(torch.tensor([[7]], device=DEV))
    warm = dec.capture()
    assert warm == 2
    oracle_step(0)
    check_state(1, oracle_step(1), "after the eager warm-up steps")
    miss_total = 0
    for r in range(3):
        i = warm + r
        dec.step()
        torch.cuda.synchronize()
        hits = oracle_step(i)
        check_state(i, hits, f"replay {r}")
        miss_total += sum(S - x for row in hits for x in row)
        l = m.num_layers - 1
        kv_len = c.sparse_end + i + 1
        q = table[i % table.shape[0]][l][0].cpu().view(1, Hq, D).contiguous()
        check_attention(f"test_captured_default_path_against_the_oracle_at_full_size[{shape}] replay {r}",
                        c.attn_out_tap[l].view(1, Hq, D).cpu().float(), q, c.k_cache_buffer[l].cpu().contiguous(),
                        c.v_cache_buffer[l].cpu().contiguous(), kv_len, 1 / math.sqrt(D), overlapped_pass_labels(c, kv_len))
    assert miss_total > 0
    assert sum(int(c.early_fetch_counts(l).sum()) for l in range(m.num_layers)) > 0       # the early fetch took part


@pytest.mark.parametrize("cfg_name,glm", [("LLAMA_3_1_8B", False), ("GLM_4_9B_1M", True)])
def test_ninety_steps_agree_across_all_selection_paths(cfg_name, glm):
    """Soak: 90 captured steps (the generated-row slack allows 96) of one prompt through the four selection paths - fused
    selection with / without the early fetch (and with the round-5 near-miss staging on top), three-launch selection with / without it - on the same query walk: sampled
    tokens of every step, the final slot -> chunk map, hit counts and both caches bit for bit.  Covers what a 6-step run does
    not: the witness-level controller of the fused selection over many steps, early-fetch
    staging slots being reused dozens of times, and the generated rows filling up."""
    from shadowkv_amd import llama
    ctx, steps = 32768, 90
    runs = {}
    for fused, early in ((True, "near"), (True, True), (True, False), (False, True), (False, False)):
        m = llama.DecoderLM(cfg=getattr(llama, cfg_name), batch_size=1, max_length=ctx, device=DEV, sparse_budget=2048, rank=160,
                            chunk_size=8, num_layers=2, seed=3, chunk_layout="inplace", overlap_attention=True)
        llama.build_synthetic_context(m, ctx, seed=11)
        m.kv_cache.fused_select = fused
        if early:
            m.kv_cache.enable_early_fetch(near=early == "near")     # ("near": + the near-miss staging in the gate/up launches, round 5)
        table = llama.make_walk_table(m, 16, seed=5)
        dec = llama.GraphDecoder(m, temperature=0.6, walk_table=table)
        dec.token.copy_(torch.tensor([[7]], device=DEV))
        warm = dec.capture()
        toks, paths, pulled = [], [], 0
        for _ in range(steps - warm):
            toks.append(int(dec.step()[0, 0]))
            if fused:
                paths += [int(p) for l in range(m.num_layers) for p in m.kv_cache.fused_select_stats(l)[:, 0]]
            if early:
                pulled += sum(int(m.kv_cache.early_fetch_counts(l).sum()) for l in range(m.num_layers))
        torch.cuda.synchronize()
        c = m.kv_cache
        assert c.fused_select_stats(0) is not None if fused else c.fused_select_stats(0) is None
        if fused:        # the carried level holds on most steps of a drifting query (the search runs in the eager warm-up steps)
            assert paths.count(0) > len(paths) // 2, (paths.count(0), len(paths))
        if early:
            assert pulled > 0
        if early == "near":      # 64 slots per head reused over ~88 steps: the staged chunks are still what the map says
            assert all(int((c.near_published_ids(l) >= 0).sum()) > 0 for l in range(m.num_layers))
        runs[(fused, early)] = (toks, c.position_ids.clone(), c._cnts_layers.clone(), c.k_cache_buffer.view(torch.int16).clone(),
                                c.v_cache_buffer.view(torch.int16).clone(), c.gen_offset)
        del dec, m
        torch.cuda.empty_cache()
    ref = runs[(False, False)]
    assert ref[5] == steps and len(set(ref[0])) > 1
    for key, run in runs.items():
        assert run[0] == ref[0], (key, "tokens")
        assert run[5] == ref[5]
        for a, b, what in zip(run[1:5], ref[1:5], ("position_ids", "cnts", "k cache", "v cache")):
            assert torch.equal(a, b), (key, what)
