"""GPU: ShadowKVCache_CPU driven exactly as LLM.layer_compute drives it (models/base.py:315-341) for
several decode steps, against a CPU mirror advanced by the oracle from the same starting state."""
import ctypes
import math

import pytest
import torch

import gen_inputs as G
import oracle
from util import attention_tolerance, check_attention, standalone_pass_labels, overlapped_pass_labels, ALPHA, assert_bits_equal, ulp_diff_bf16, rope_pair_bound, record_parity, REBUILD_FLIP_BOUND

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(case):
    from shadowkv_amd.kv_cache import ShadowKVCache_CPU
    c, inp = G.CASES[case], G.make_inputs(case)
    cache = ShadowKVCache_CPU(G.config_of(case), batch_size=1, max_length=c["L"], device=DEV, dtype=torch.bfloat16,
                              sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"])
    k_pre = inp["k_pre"].to(DEV)
    cache.get_svd(k_pre, 0)
    k_roped = G.rope_torch(case, inp["k_pre"], inp["cos_sin"], torch.arange(c["L"]).unsqueeze(0)).to(DEV)
    cache.prefill_kv_cache(inp["v"].to(DEV), 0, k_roped, inp["q_last"].to(DEV))
    cache.H2D()
    torch.cuda.synchronize()
    # the V table is page-locked memory of the HIP runtime torch itself uses (skv_host_alloc): torch sees it as pinned, so
    # copies into it are asynchronous DMA; a slice keeps the allocation alive after the cache is gone
    assert cache.v_cache_cpu.is_pinned() and not cache.v_cache_cpu.is_cuda
    return cache, c, inp


def test_update_kv_cache_native_launch_matches_the_sliced_copies():
    """update_kv_cache (kv_cache.py:1227-1271) as ONE native launch: the new K rows (contiguous) and V rows (a strided view of
    the fused projection, as the host model hands them over) land in rows [sparse_end + gen, ...) of both buffers, everything
    else keeps its bytes; rows past the end of the buffer are dropped like the reference's zero-length slice."""
    cache, c, inp = _build("llama_small")
    kv, D = c["kv_heads"], c["head_dim"]
    g = torch.Generator(device=DEV).manual_seed(4)
    rows = cache.k_cache_buffer.shape[-2]
    slack = rows - cache.sparse_end
    for incoming, gen in ((1, 0), (2, 5), (3, slack - 1)):
        cache.gen_offset = gen
        k0, v0 = cache.k_cache_buffer[0].clone(), cache.v_cache_buffer[0].clone()
        fused = torch.randn(1, incoming, (32 + 2 * kv) * D, device=DEV, generator=g).bfloat16()
        k_new = torch.randn(1, kv, incoming, D, device=DEV, generator=g).bfloat16()
        v_new = fused[..., (32 + kv) * D:].view(1, incoming, kv, D).transpose(1, 2)          # [1, kv, incoming, D], strided
        assert incoming == 1 or not v_new.is_contiguous()
        cache.update_kv_cache(k_new, v_new, 0)
        torch.cuda.synchronize()
        lo = cache.sparse_end + gen
        n = min(incoming, rows - lo)
        k0[:, :, lo:lo + n] = k_new[:, :, :n]
        v0[:, :, lo:lo + n] = v_new[:, :, :n]
        assert_bits_equal(cache.k_cache_buffer[0], k0, f"K buffer, incoming {incoming} at row {lo}")
        assert_bits_equal(cache.v_cache_buffer[0], v0, f"V buffer, incoming {incoming} at row {lo}")
        assert cache.gen_offset == gen + incoming              # (the fixture cache has one layer: the last one advances)


def test_pinned_v_table_outlives_the_cache_through_its_views():
    import gc
    cache, c, inp = _build("llama_small")
    view = cache.v_cache_cpu[0][0, 1]
    want = view.clone()
    del cache
    gc.collect()
    assert view.is_pinned() and torch.equal(view, want)


@pytest.mark.parametrize("case", ["llama_small", "llama_cpu_b1024", "glm_small"])
def test_decode_steps_match_oracle(case):
    from shadowkv_amd import tensor_op
    cache, c, inp = _build(case)
    kv, Hq, D, C, S = c["kv_heads"], c["q_heads"], c["head_dim"], c["chunk"], cache.select_sets
    Gq = Hq // kv
    cs_dev = inp["cos_sin"].to(DEV)
    # CPU mirror of the state the GPU starts from
    lm = cache.k_landmark[0][0].cpu().contiguous(); lm_idx = cache.k_landmark_idx[0][0].cpu().contiguous()
    N = lm.shape[1]; T = (N + 255) // 256
    U = cache.U[0].cpu().contiguous(); SV = cache.SV[0].cpu().contiguous()
    pos = cache.position_ids[0][0].cpu().clone()
    kbuf = cache.k_cache_buffer[0][0].cpu().clone(); vbuf = cache.v_cache_buffer[0][0].cpu().clone()
    vhost = cache.v_cache_cpu[0][0].clone()
    rows = kbuf.shape[1]
    hit_rates = []
    for t in range(inp["q_steps"].shape[0]):
        q = inp["q_steps"][t]                                           # [1, Hq, 1, D]
        # ---------------- GPU, in the reference's call order with its two-stream overlap
        qd = q.to(DEV)
        knew = torch.randn(1, kv, 1, D, generator=torch.Generator().manual_seed(t)).bfloat16()
        vnew = torch.randn(1, kv, 1, D, generator=torch.Generator().manual_seed(100 + t)).bfloat16()
        cache.update_kv_cache(knew.to(DEV), vnew.to(DEV), 0)           # single layer == last layer: offsets advance
        position_ids = cache.get_retrieval_position_ids(layer_idx=0, query_states=qd)
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(cache.copy_stream):
            cache.copy_stream.wait_stream(cur)
            v_view = cache.get_value_cache(0, position_ids)
        k_view = cache.get_key_cache(layer_idx=0, position_ids=position_ids, rope_func=None, cos_sin_cache=cs_dev)
        cur.wait_stream(cache.copy_stream)
        attn = tensor_op.sparse_attention_decode(qd, k_view, v_view)
        torch.cuda.synchronize()
        # ---------------- CPU mirror through the oracle
        gen_row = cache.sparse_end + t
        kbuf[:, gen_row] = knew[0, :, 0]; vbuf[:, gen_row] = vnew[0, :, 0]
        Dm = torch.zeros(kv, Gq, N, dtype=torch.bfloat16); P = torch.zeros_like(Dm)
        oracle.batch_gemm_softmax(q.view(kv, Gq, D).contiguous(), lm, Dm, torch.zeros(kv, T, Gq), torch.zeros(kv, T, Gq),
                                  P, kv, Gq, N, D, ALPHA)
        sel = oracle.group_max_topk(P, lm_idx, kv, Gq, N, S)
        off = torch.zeros(kv, S, dtype=torch.int32); cnt = torch.zeros(kv, dtype=torch.int32)
        oracle.reorder_keys_and_compute_offsets(pos, sel, off, cnt, 1, kv, S)
        assert torch.equal(pos, cache.position_ids[0][0].cpu()), f"step {t}: position_ids"
        assert torch.equal(off.flatten(), cache.offsets.cpu()), f"step {t}: offsets"
        assert torch.equal(cnt, cache.cnts.cpu()), f"step {t}: cnts"
        hit_rates.append(float(cnt.sum()) / (kv * S))
        oracle.gather_copy_with_offsets(vhost, vbuf, None, off, cnt, None, 1, kv, vhost.stride(0), S * C * D,
                                        cache.sparse_start * D, rows * D, S)
        assert_bits_equal(vbuf, cache.v_cache_buffer[0][0], f"step {t}: V buffer")
        oracle.gather_copy_d2d_with_offsets(kbuf, off, cnt, 1, kv, S * C * D, cache.sparse_start * D, rows * D, S)
        pre = torch.zeros(1, kv, S * C, D, dtype=torch.bfloat16)
        ids32 = pos.to(torch.int32).view(1, kv, S).contiguous()
        oracle.batch_gather_gemm(U, SV, None, None, ids32, pre, 1, kv, U.shape[1], D, c["rank"], S * C, 0, C, cnt)
        kb4 = kbuf.view(1, kv, rows, D)
        cs = inp["cos_sin"]
        ints = (1, kv, S * C, D, pre.stride(0), pre.stride(1), pre.stride(2), 1, cs.stride(0), ids32.stride(0),
                ids32.stride(1), ids32.stride(2), kb4.stride(0), kb4.stride(1), kb4.stride(2), cache.sparse_start,
                cache.sparse_end, 64, C)
        (oracle.apply_rotary_pos_emb_push_cache_opt_glm if c["glm"] else oracle.apply_rotary_pos_emb_push_cache_opt)(
            pre, cs, ids32, kb4, cnt, *ints)
        kgpu = cache.k_cache_buffer[0][0].cpu()
        d = ulp_diff_bf16(kbuf, kgpu)
        frac, _ = record_parity(f"test_decode_steps_match_oracle[{case}] step {t}", d[:, cache.sparse_start:cache.sparse_end],
                                "post-RoPE")
        assert frac < REBUILD_FLIP_BOUND, f"step {t}: K buffer"
        # rebuilt rows: |device - oracle| <= 2^-6 (|x1| + |x2|) per rotation pair (x = the oracle's pre-RoPE row); rows
        # the oracle did not rebuild (hits: pre == 0) must be identical
        kdiff = (kbuf.float() - kgpu.float()).abs()[:, cache.sparse_start:cache.sparse_end]
        bound = rope_pair_bound(pre[0], c["glm"])
        assert bool((kdiff <= bound).all()), f"step {t}: K rows exceed the one-ulp-flip bound by {float((kdiff - bound).max())}"
        # hit rows and everything outside the rebuilt range must be bit-identical
        for h in range(kv):
            r0 = cache.sparse_start + int(cnt[h]) * C
            assert_bits_equal(kbuf[h, :r0], kgpu[h, :r0], f"step {t}: K rows below the rebuilt range")
            assert_bits_equal(kbuf[h, cache.sparse_end:], kgpu[h, cache.sparse_end:], f"step {t}: generated rows")
        kbuf.copy_(kgpu)                                                # keep the mirror on the device's K bits
        # returned views: [:, :, :sparse_end + gen]
        assert k_view.shape[2] == cache.sparse_end + t + 1 and v_view.shape[2] == k_view.shape[2]
        # (standalone pass: bf16 softmax weights on the matrix pipe - the oracle rounds them at the same points)
        n_att = k_view.shape[2]
        check_attention(f"test_decode_steps_match_oracle[{case}] step {t}", attn.view(1, Hq, D).cpu().float(),
                        q.view(1, Hq, D).contiguous(), kgpu.unsqueeze(0).contiguous(), vbuf.unsqueeze(0).contiguous(), n_att,
                        1 / math.sqrt(D), standalone_pass_labels(1, Hq, kv, n_att, tensor_op.default_attention_splits(1, kv, n_att)))
    assert max(hit_rates) > 0.0   # the random-walk queries do re-select resident chunks


def test_legacy_two_launch_key_path_equals_fused():
    """tensor_op.batch_gather_gemm_rotary_pos_emb_cuda (reference signature, two launches through an
    `output` buffer) and the fused rebuild must write identical bits: same MFMA order, same roundings."""
    from shadowkv_amd import tensor_op
    cache, c, inp = _build("llama_small")
    kv, D, C, S = c["kv_heads"], c["head_dim"], c["chunk"], cache.select_sets
    cs = inp["cos_sin"].to(DEV)
    cache.update_kv_cache(torch.zeros(1, kv, 1, D, dtype=torch.bfloat16, device=DEV),
                          torch.zeros(1, kv, 1, D, dtype=torch.bfloat16, device=DEV), 0)
    pos = cache.get_retrieval_position_ids(layer_idx=0, query_states=inp["q_steps"][0].to(DEV))
    k1 = cache.k_cache_buffer[0].clone(); k2 = cache.k_cache_buffer[0].clone()
    tensor_op.rebuild_keys(cache.U[0], cache.SV[0], cs, pos, cache.cnts, k1, cache.sparse_start, C)
    out = torch.zeros(1, kv, S * C, D, dtype=torch.bfloat16, device=DEV)
    tensor_op.batch_gather_gemm_rotary_pos_emb_cuda(cache.U[0], cache.SV[0], cs, pos, out, C, k2, cache.sparse_start,
                                                    cache.sparse_end, cache.cnts)
    torch.cuda.synchronize()
    assert_bits_equal(k1, k2, "fused vs two-launch K rebuild")


@pytest.mark.parametrize("case", ["llama_small", "glm_small"])
def test_fetch_kv_single_launch_equals_two_calls(case):
    """cache.fetch_kv (K rebuild tiles + V landing blocks in one grid) must leave the same bytes in both caches
    as get_value_cache + get_key_cache."""
    ca, c, inp = _build(case)
    cb, _, _ = _build(case)
    cs = inp["cos_sin"].to(DEV)
    for t in range(3):
        q = inp["q_steps"][t].to(DEV)
        ida = ca.get_retrieval_position_ids(layer_idx=0, query_states=q)
        idb = cb.get_retrieval_position_ids(layer_idx=0, query_states=q)
        ca.get_value_cache(0, ida)
        ca.get_key_cache(layer_idx=0, position_ids=ida, rope_func=None, cos_sin_cache=cs)
        cb.fetch_kv(0, idb, cs)
        torch.cuda.synchronize()
        assert torch.equal(ca.position_ids, cb.position_ids)
        assert_bits_equal(ca.k_cache_buffer, cb.k_cache_buffer, f"step {t}: K")
        assert_bits_equal(ca.v_cache_buffer, cb.v_cache_buffer, f"step {t}: V")


@pytest.mark.parametrize("case", ["llama_small", "llama_cpu_b1024", "glm_small"])
def test_inplace_layout_same_set_same_rows_no_hit_moves(case):
    """select_fetch_inplace against the reference-order path (get_retrieval_position_ids + fetch_kv) from the same
    starting state over several steps: the same chunk SET per head and the same hit count every step; slot i of the
    in-place cache holds chunk position_ids[i] with exactly the K / V rows the reference-order cache holds for that
    chunk; a chunk selected again never changes slot; attention over both layouts agrees to summation order."""
    from shadowkv_amd import tensor_op
    ref, c, inp = _build(case)
    inp_cache, _, _ = _build(case)
    kv, D, C, S = c["kv_heads"], c["head_dim"], c["chunk"], ref.select_sets
    cs_dev = inp["cos_sin"].to(DEV)
    ss, se = ref.sparse_start, ref.sparse_end
    assert torch.equal(ref.position_ids, inp_cache.position_ids)
    for t in range(inp["q_steps"].shape[0]):
        qd = inp["q_steps"][t].to(DEV)
        before = inp_cache.position_ids[0][0].clone()
        ids = ref.get_retrieval_position_ids(layer_idx=0, query_states=qd)
        ref.fetch_kv(0, ids, cs_dev)
        inp_cache.select_fetch_inplace(0, qd, cs_dev)
        torch.cuda.synchronize()
        a, b = ref.position_ids[0][0], inp_cache.position_ids[0][0]
        assert torch.equal(a.sort(dim=-1).values, b.sort(dim=-1).values), t
        assert torch.equal(ref.cnts, inp_cache.cnts)
        for h in range(kv):
            keep = before[h] == b[h]                                      # slots whose chunk survived
            assert int(keep.sum()) == int(inp_cache.cnts[h]), (t, h)
            # misses: ascending ids into ascending freed slots
            new_ids = b[h][~keep]
            assert torch.equal(new_ids, new_ids.sort().values)
            # row contents per chunk id, both layouts
            order_a, order_b = a[h].argsort(), b[h].argsort()
            for buf_a, buf_b in ((ref.k_cache_buffer, inp_cache.k_cache_buffer), (ref.v_cache_buffer, inp_cache.v_cache_buffer)):
                ra = buf_a[0][0, h, ss:se].view(S, C * D)[order_a]
                rb = buf_b[0][0, h, ss:se].view(S, C * D)[order_b]
                assert_bits_equal(rb, ra, f"step {t} head {h}")
            want_v = inp["v"][0, h].view(-1, C, D)[b[h].cpu()].reshape(-1, D)
            assert_bits_equal(inp_cache.v_cache_buffer[0][0, h, ss:se], want_v)
        o_ref = tensor_op.sparse_attention_decode(qd, ref.k_cache_buffer[0][:, :, :se], ref.v_cache_buffer[0][:, :, :se])
        o_inp = tensor_op.sparse_attention_decode(qd, inp_cache.k_cache_buffer[0][:, :, :se], inp_cache.v_cache_buffer[0][:, :, :se])
        assert torch.allclose(o_ref.float(), o_inp.float(), rtol=2 ** -7, atol=2e-3)


def test_v_table_in_hbm_gives_the_same_cache_bytes():
    """v_offload=False (V table resident in HBM, like the reference's GPU-only ShadowKVCache) through the same
    kernels: identical buffers and ids after several steps."""
    from shadowkv_amd.kv_cache import ShadowKVCache_CPU
    case = "llama_cpu_b1024"
    a, c, inp = _build(case)
    b = ShadowKVCache_CPU(G.config_of(case), batch_size=1, max_length=c["L"], device=DEV, dtype=torch.bfloat16,
                          sparse_budget=c["budget"], chunk_size=c["chunk"], rank=c["rank"], v_offload=False)
    assert b.v_cache_cpu.is_cuda and not a.v_cache_cpu.is_cuda
    b.get_svd(inp["k_pre"].to(DEV), 0)
    k_roped = G.rope_torch(case, inp["k_pre"], inp["cos_sin"], torch.arange(c["L"]).unsqueeze(0)).to(DEV)
    b.prefill_kv_cache(inp["v"].to(DEV), 0, k_roped, inp["q_last"].to(DEV))
    b.H2D()
    cs_dev = inp["cos_sin"].to(DEV)
    for t in range(inp["q_steps"].shape[0]):
        qd = inp["q_steps"][t].to(DEV)
        for cache in (a, b):
            ids = cache.get_retrieval_position_ids(layer_idx=0, query_states=qd)
            cache.fetch_kv(0, ids, cs_dev)
    torch.cuda.synchronize()
    assert torch.equal(a.position_ids, b.position_ids)
    assert_bits_equal(a.v_cache_buffer, b.v_cache_buffer)
    assert_bits_equal(a.k_cache_buffer, b.k_cache_buffer)


@pytest.mark.parametrize("case", ["llama_cpu_b1024", "glm_small"])
def test_overlapped_attention_equals_fetch_then_attend(case):
    """select_fetch_attend_inplace (attention over the resident rows inside the fetch launch, the miss tiles attended by
    the workgroups that build them, merge kernel) against select_fetch_inplace + sparse_attention_decode from the same state, several steps: identical
    cache bytes / ids, attention outputs equal up to the order of the f32 sums, and both within the oracle's f64
    attention tolerance."""
    from shadowkv_amd import tensor_op
    a, c, inp = _build(case)
    b, _, _ = _build(case)
    assert b.can_overlap_attention()
    cs_dev = inp["cos_sin"].to(DEV)
    kv, Hq, D = c["kv_heads"], c["q_heads"], c["head_dim"]
    for t in range(inp["q_steps"].shape[0]):
        qd = inp["q_steps"][t].to(DEV)
        knew = torch.randn(1, kv, 1, D, generator=torch.Generator().manual_seed(t)).bfloat16().to(DEV)
        vnew = torch.randn(1, kv, 1, D, generator=torch.Generator().manual_seed(50 + t)).bfloat16().to(DEV)
        for cache in (a, b):
            cache.update_kv_cache(knew, vnew, 0)
        rows = a.sparse_end + a.gen_offset
        a.select_fetch_inplace(0, qd, cs_dev)
        o_ref = tensor_op.sparse_attention_decode(qd, a.k_cache_buffer[0], a.v_cache_buffer[0], kv_len=rows)
        kvd = torch.tensor([rows], dtype=torch.int32, device=DEV)
        o_new = b.select_fetch_attend_inplace(0, qd, cs_dev, kv_len=0, kv_len_dev=kvd) if t % 2 else \
            b.select_fetch_attend_inplace(0, qd, cs_dev, kv_len=rows)
        torch.cuda.synchronize()
        assert torch.equal(a.position_ids, b.position_ids) and torch.equal(a.cnts, b.cnts)
        assert_bits_equal(a.k_cache_buffer, b.k_cache_buffer)
        assert_bits_equal(a.v_cache_buffer, b.v_cache_buffer)
        assert o_new.shape == o_ref.shape
        # both HIP paths against the oracle's f32 output at the north-star bound (1e-3 relative + half a bf16 ulp of the
        # output rounding), the bound of the standalone kernel's test
        # (both paths round softmax weights to bf16 for the matrix pipe: the standalone pass all of them against its waves'
        # running maxima, the overlapped path those of the miss tiles against the tile maximum - each has its own oracle labels)
        kc, vc = a.k_cache_buffer[0].cpu(), a.v_cache_buffer[0].cpu()
        buf_rows = kc.shape[2]
        for name, o, labels in (("overlapped", o_new, overlapped_pass_labels(b, rows)),
                                ("fetch-then-attend", o_ref, standalone_pass_labels(1, Hq, kv, rows, tensor_op.default_attention_splits(1, kv, buf_rows)))):
            check_attention(f"test_overlapped_attention_equals_fetch_then_attend[{case}] step {t} {name}",
                            o.cpu().float().view(1, Hq, D), qd.cpu().view(1, Hq, D), kc, vc, rows, 1.0 / math.sqrt(D), labels)


def _headline_cache(kv_heads, glm, L=8192, seed=11, resident_sets=None, max_length=None, budget=2048):
    """ShadowKVCache_CPU with the headline layout (budget 2,048 -> S = 256, 48 outlier chunks, sparse region rows
    [448, 2496), 96 generated rows) over an L-token synthetic context whose keys are exactly rank 160.  budget 1,024 / 4,096:
    the reference's 60K / 244K regimes (test/e2e.py:35-116): S = 128 / 512, 24 / 96 outlier chunks."""
    from shadowkv_amd import llama
    from shadowkv_amd.kv_cache import ShadowKVCache_CPU
    mc = llama.ModelConfig(num_hidden_layers=1, num_key_value_heads=kv_heads, rope_style="glm" if glm else "neox",
                           rope_theta=10000.0 if glm else 500000.0)
    cache = ShadowKVCache_CPU(mc, batch_size=1, max_length=max_length or L, device=DEV, dtype=torch.bfloat16, sparse_budget=budget,
                              chunk_size=8, rank=160, resident_sets=resident_sets)
    cs, g = _headline_prefill(cache, L, seed)
    return cache, cs, g


def _headline_prefill(cache, L, seed):
    """Prefills `cache` (one layer) with an L-token synthetic context (L <= cache.max_length); returns (cos_sin, generator)."""
    from shadowkv_amd import llama, tensor_op
    mc, kv_heads = cache.config, cache.num_key_value_heads
    glm = mc.rope_style == "glm"
    g = torch.Generator(device=DEV).manual_seed(seed)
    D, r = 128, 160
    cs = llama.build_cos_sin_cache(mc, cache.max_length + 256, torch.device(DEV), torch.bfloat16)
    U = torch.randn(1, L, r, device=DEV, generator=g).bfloat16()
    SV = (torch.randn(1, kv_heads, D, r, device=DEV, generator=g) / math.sqrt(r)).bfloat16()
    cache.U = U.unsqueeze(0).contiguous(); cache.SV = SV.unsqueeze(0).contiguous()
    k_pre = torch.einsum("blr,bhdr->bhld", U.float(), SV.float()).bfloat16().contiguous()
    pos = torch.arange(L, device=DEV).unsqueeze(0)
    if glm:
        c, s_ = cs[pos].unsqueeze(1)[..., :32], cs[pos].unsqueeze(1)[..., 32:]
        xe, xo = k_pre[..., 0:64:2], k_pre[..., 1:64:2]
        rot = torch.stack((xe * c - xo * s_, xo * c + xe * s_), dim=-1).flatten(-2)
        k_roped = torch.cat((rot, k_pre[..., 64:]), dim=-1).contiguous()
    else:
        k_roped = tensor_op.apply_rotary_pos_emb_cuda(k_pre, cs, pos.unsqueeze(1).expand(-1, kv_heads, -1).contiguous())
    v = torch.randn(1, kv_heads, L, D, device=DEV, generator=g).bfloat16()
    q_last = (torch.randn(1, 32, 1, D, device=DEV, generator=g) * 1.5).bfloat16()
    cache.prefill_kv_cache(v, 0, k_roped, q_last)
    cache.H2D()
    return cs, g


@pytest.mark.parametrize("kv_heads,glm,hit,ctx,budget", [
    (8, False, 0.67, 8192, 2048), (8, False, 0.0, 8192, 2048), (8, False, 1.0, 8192, 2048),
    (4, False, 0.67, 8192, 2048), (4, True, 0.0, 8192, 2048), (4, True, 0.67, 8192, 2048),
    # BASELINE.json configs 2 and 3 at their full per-layer size
    (8, False, 0.67, 131072, 2048), (4, True, 0.67, 204800, 2048),
    # the reference's other two regimes (test/e2e.py:35-116): budget 4096 (S = 512: 64 miss tiles + 24 splits = 88 attention
    # records per head) incl. one layer at the full 244K size, G = 4 and G = 8 (Yi-9B / GLM-4); budget 1024 (S = 128) at 60K
    (8, False, 0.6, 32768, 4096), (8, False, 0.0, 32768, 4096), (4, False, 0.6, 32768, 4096), (4, True, 0.6, 32768, 4096),
    (8, False, 0.6, 249856, 4096), (8, False, 0.6, 61440, 1024), (4, True, 0.0, 16384, 1024)])
def test_overlapped_attention_at_headline_shape_against_f32_oracle(kv_heads, glm, hit, ctx, budget):
    """The path the headline number runs on (bench.py defaults: in-place layout, attention over the resident rows inside
    the fetch launch, miss tiles attended where they are built, merge kernel) at the headline shape: S = 256 chunks, sparse region [448, 2496),
    kv_len = 2,499, G = 4 and G = 8, chunk hit rates 0 / 0.67 / 1; also one layer of the 131,072-token Llama-3-1048K and of
    the 204,800-token GLM-4 configuration (BASELINE.json configs 2 and 3) at full size.  Checked against the oracle: selected set bit-exact,
    V rows byte-exact against the host table, K rows within the one-ulp-flip bound of the oracle's rebuild, and the
    attention output against the oracle's F32 result over the device's K / V bytes at 1e-3 |ref| + half a bf16 ulp."""
    cache, cs, g = _headline_cache(kv_heads, glm, L=ctx, budget=budget)
    Hq, D, C, S = 32, 128, 8, cache.select_sets
    Gq = Hq // kv_heads
    n_out = budget // 1024 * 24                                              # outlier chunks (kv_cache.py:548)
    assert S == budget // 8 and cache.sparse_start == 64 + 8 * n_out and cache.sparse_end == cache.sparse_start + budget
    assert cache.k_cache_buffer.shape[-2] == cache.sparse_end + 96 and cache.can_overlap_attention()
    assert (S, cache.sparse_start, cache.sparse_end) == {2048: (256, 448, 2496), 4096: (512, 832, 4928), 1024: (128, 256, 1280)}[budget]
    q = (torch.randn(1, Hq, 1, D, device=DEV, generator=g) * 1.5).bfloat16()
    gen = 3
    for buf in (cache.k_cache_buffer, cache.v_cache_buffer):
        buf[0][:, :, cache.sparse_end:cache.sparse_end + gen] = torch.randn(1, kv_heads, gen, D, device=DEV, generator=g).bfloat16()
    kv_len = cache.sparse_end + gen
    cache.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)              # step 1: the region now holds top-S(q)
    torch.cuda.synchronize()
    lm_idx = cache.k_landmark_idx[0][0].cpu()
    pos = cache.position_ids[0][0].cpu().clone()
    n_replace = S - int(round(hit * S))
    gc = torch.Generator().manual_seed(3)
    for h in range(kv_heads):                                               # evict n_replace of the selected chunks
        pool = torch.tensor(sorted(set(lm_idx[h].tolist()) - set(pos[h].tolist())))
        slots = torch.randperm(S, generator=gc)[:n_replace]
        pos[h, slots] = pool[torch.randperm(len(pool), generator=gc)[:n_replace]]
    cache.position_ids[0][0].copy_(pos.to(DEV))
    out = cache.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)         # step 2: same query -> n_replace misses
    torch.cuda.synchronize()
    assert cache.cnts.cpu().tolist() == [S - n_replace] * kv_heads
    # the selected set is the oracle's
    lm = cache.k_landmark[0][0].cpu().contiguous(); N = lm.shape[1]; T = (N + 255) // 256
    Dm = torch.zeros(kv_heads, Gq, N, dtype=torch.bfloat16); P = torch.zeros_like(Dm)
    oracle.batch_gemm_softmax(q.cpu().view(kv_heads, Gq, D).contiguous(), lm, Dm, torch.zeros(kv_heads, T, Gq),
                              torch.zeros(kv_heads, T, Gq), P, kv_heads, Gq, N, D, ALPHA)
    sel = oracle.group_max_topk(P, lm_idx.contiguous(), kv_heads, Gq, N, S)
    ids = cache.position_ids[0][0].cpu()
    assert torch.equal(ids.sort(dim=-1).values, sel.sort(dim=-1).values)
    # V rows: byte-exact copies of the host table's chunks; K rows: the oracle's rebuild within the one-ulp-flip bound
    kbuf, vbuf = cache.k_cache_buffer[0].cpu(), cache.v_cache_buffer[0].cpu()
    vhost = cache.v_cache_cpu[0][0]
    want_v = torch.stack([vhost[h][ids[h]] for h in range(kv_heads)]).view(kv_heads, S * C, D)
    assert_bits_equal(vbuf[0][:, cache.sparse_start:cache.sparse_end], want_v, "V rows of the sparse region")
    pre = torch.zeros(1, kv_heads, S * C, D, dtype=torch.bfloat16)
    ids32 = ids.to(torch.int32).view(1, kv_heads, S).contiguous()
    zero = torch.zeros(kv_heads, dtype=torch.int32)
    U, SV, csc = cache.U[0].cpu().contiguous(), cache.SV[0].cpu().contiguous(), cs.cpu()
    oracle.batch_gather_gemm(U, SV, None, None, ids32, pre, 1, kv_heads, U.shape[1], D, 160, S * C, 0, C, zero)
    kor = kbuf.clone()
    ints = (1, kv_heads, S * C, D, pre.stride(0), pre.stride(1), pre.stride(2), 1, csc.stride(0), ids32.stride(0),
            ids32.stride(1), ids32.stride(2), kor.stride(0), kor.stride(1), kor.stride(2), cache.sparse_start,
            cache.sparse_end, 64, C)
    (oracle.apply_rotary_pos_emb_push_cache_opt_glm if glm else oracle.apply_rotary_pos_emb_push_cache_opt)(
        pre, csc, ids32, kor, zero, *ints)
    kd = (kor.float() - kbuf.float()).abs()[0][:, cache.sparse_start:cache.sparse_end]
    record_parity(f"test_overlapped_attention_at_headline_shape[kv{kv_heads}-glm{int(glm)}-hit{hit}-ctx{ctx}-b{budget}]",
                  ulp_diff_bf16(kor[0][:, cache.sparse_start:cache.sparse_end], kbuf[0][:, cache.sparse_start:cache.sparse_end]),
                  "post-RoPE")     # (hit rows were placed by the prefill's torch einsum, miss rows by the rebuild kernel)
    bound = rope_pair_bound(pre[0], glm)
    assert bool((kd <= bound).all()), f"K rows exceed the bound by {float((kd - bound).max())}"
    # attention: F32 oracle over the device's own K / V bytes
    # (the miss tiles' P.V runs on the MFMA with bf16 weights, like flash-attn's: the oracle rounds the same weights against
    # the same tile maxima, from the slots the selection assigned)
    qc = q.cpu().view(1, Hq, D).contiguous()
    a32 = check_attention(f"test_overlapped_attention_at_headline_shape[kv{kv_heads}-glm{int(glm)}-hit{hit}-ctx{ctx}-b{budget}]",
                          out.view(1, Hq, D).cpu().float(), qc, kbuf, vbuf, kv_len, 1 / math.sqrt(D), overlapped_pass_labels(cache, kv_len))
    _, aabs = oracle.sparse_attention(qc, kbuf, vbuf.abs(), kv_len, 1 / math.sqrt(D))
    tol = attention_tolerance(a32, aabs)
    # kv_len from device memory (the state is now all hits: same rows, another split of the f32 sums); a host kv_len
    # past the buffer is refused
    kvd = torch.tensor([kv_len], dtype=torch.int32, device=DEV)
    out2 = cache.select_fetch_attend_inplace(0, q, cs, kv_len=0, kv_len_dev=kvd)
    torch.cuda.synchronize()
    assert cache.cnts.cpu().tolist() == [S] * kv_heads
    check_attention(f"test_overlapped_attention_at_headline_shape[kv{kv_heads}-glm{int(glm)}-hit{hit}-ctx{ctx}-b{budget}] all hits",
                    out2.view(1, Hq, D).cpu().float(), qc, kbuf, vbuf, kv_len, 1 / math.sqrt(D), overlapped_pass_labels(cache, kv_len))
    assert int((overlapped_pass_labels(cache, kv_len)[0] >= 0).sum()) == 0      # (no miss tile: every weight stays f32)
    with pytest.raises(ValueError):
        cache.select_fetch_attend_inplace(0, q, cs, kv_len=cache.k_cache_buffer.shape[-2] + 1)


@pytest.mark.parametrize("kv_heads,glm,overlap", [(8, False, True), (8, False, False), (4, True, True), (8, False, "near")])
def test_resident_set_of_512_chunks_attends_exactly_the_selection(kv_heads, glm, overlap):
    """resident_sets = 512 > select_sets = 256 (in-place layout): over a walk of queries, step by step against a cache
    with the reference's resident set (256) fed the same queries - the selected sets are identical, the hit counts are
    never lower, every occupied slot holds its chunk's V rows byte for byte (also the slots that are resident but not
    selected), and the attention output equals the oracle's F32 attention over [local + outliers | the selected chunks |
    generated rows] of the device's own K / V bytes at 1e-3 |ref| + half a bf16 ulp - with the attention inside the fetch
    launch (overlap) and with the standalone pass over the slot list."""
    from shadowkv_amd import tensor_op
    R = 512
    big, cs, g = _headline_cache(kv_heads, glm, resident_sets=R)
    ref, _, _ = _headline_cache(kv_heads, glm)
    near = overlap == "near"     # (round 5: + the early fetch and the near-miss staging on the 512-slot cache - a chunk staged ahead
    #                              may be resident but not selected there; what is attended and every slot's bytes must not change)
    if near:
        big.near_lists = 2
        big.enable_early_fetch(near=True)
        gw = torch.Generator(device=DEV).manual_seed(9)
    Hq, D, C, S = 32, 128, 8, big.select_sets
    assert big.sparse_end == 448 + R * C and big.k_cache_buffer.shape[-2] == 448 + R * C + 96 and ref.sparse_end == 2496
    gen = 2
    newk = torch.randn(1, kv_heads, gen, D, device=DEV, generator=g).bfloat16()
    newv = torch.randn(1, kv_heads, gen, D, device=DEV, generator=g).bfloat16()
    for c in (big, ref):
        c.k_cache_buffer[0][:, :, c.sparse_end:c.sparse_end + gen] = newk
        c.v_cache_buffer[0][:, :, c.sparse_end:c.sparse_end + gen] = newv
    q32 = torch.randn(1, Hq, 1, D, device=DEV, generator=g) * 1.5
    vhost = big.v_cache_cpu[0][0]
    more_hits = 0
    for step in range(6):
        q32 = q32 + 0.45 * torch.randn(q32.shape, device=DEV, generator=g)
        q = q32.bfloat16()
        outs = []
        for c in (big, ref):
            kv_len = c.sparse_end + gen
            if overlap:
                outs.append(c.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len))
            else:
                c.select_fetch_inplace(0, q, cs)
                outs.append(tensor_op.sparse_attention_decode(q, c.k_cache_buffer[0], c.v_cache_buffer[0], kv_len=kv_len,
                                                              **c.attend_slot_args()))
        torch.cuda.synchronize()
        if near:
            _check_staging_invariant(big, 0)
            _gate_up_with_near_pull(big, 0, gw)
            _check_staging_invariant(big, 0)
        slots = big._dst_slots.view(kv_heads, S).cpu().long()                # the S attended slots per head
        ids_big = big.position_ids[0][0].cpu()
        sel_big = torch.gather(ids_big, 1, slots).sort(dim=-1).values
        assert torch.equal(sel_big, ref.position_ids[0][0].cpu().sort(dim=-1).values), f"step {step}: selected sets differ"
        hb, hr = big._cnts_layers[0].cpu(), ref._cnts_layers[0].cpu()
        assert bool((hb >= hr).all()), (step, hb.tolist(), hr.tolist())
        more_hits += int((hb - hr).sum())
        vbuf = big.v_cache_buffer[0].cpu()
        for h in range(kv_heads):                                            # every occupied slot: its chunk's V rows
            occ = (ids_big[h] >= 0).nonzero().flatten()
            want = vhost[h][ids_big[h][occ]].view(len(occ), C, D)
            got = vbuf[0, h, big.sparse_start:big.sparse_end].view(R, C, D)[occ]
            assert_bits_equal(got, want, f"step {step} head {h}: V rows of the resident slots")
        for name, c, o in (("512 resident", big, outs[0]), ("256 resident", ref, outs[1])):
            # F32 oracle over this cache's own K / V bytes: [local + outliers | the S attended chunks | generated rows]
            sl = c._dst_slots.view(kv_heads, S).cpu().long()
            rows = (c.sparse_start + sl.unsqueeze(-1) * C + torch.arange(C)).view(kv_heads, S * C)
            kb, vb = c.k_cache_buffer[0].cpu(), c.v_cache_buffer[0].cpu()

            def view_of(buf):
                sel_rows = torch.gather(buf[0], 1, rows.unsqueeze(-1).expand(-1, -1, D))
                return torch.cat([buf[0][:, :c.sparse_start], sel_rows, buf[0][:, c.sparse_end:c.sparse_end + gen]],
                                 dim=1).unsqueeze(0).contiguous()
            kview, vview = view_of(kb), view_of(vb)
            n_att = kview.shape[2]
            if overlap:          # miss tiles (virtual slots j >= cnt, tile j // 8) round their weights against the tile maximum
                cn = c.cnts.cpu()
                grp = torch.full((1, kv_heads, n_att), -1, dtype=torch.int32)
                for h in range(kv_heads):
                    j = torch.arange(int(cn[h]), S)
                    r = (c.sparse_start + j.unsqueeze(-1) * C + torch.arange(C)).view(-1)
                    grp[0, h, r] = (j // 8).to(torch.int32).repeat_interleave(C)
                labels = (grp, torch.zeros_like(grp))
            elif c.resident_sets != S:
                labels = None    # standalone pass over a slot list: the VALU body, f32 weights
            else:                # standalone all-MFMA pass over the buffer's own row order
                kview, vview = kb[:, :, :c.sparse_end + gen].contiguous(), vb[:, :, :c.sparse_end + gen].contiguous()
                n_att = kview.shape[2]
                labels = standalone_pass_labels(1, Hq, kv_heads, n_att, tensor_op.default_attention_splits(1, kv_heads, kb.shape[2]))
            check_attention(f"test_resident_set_of_512_chunks[kv{kv_heads}-glm{int(glm)}-overlap{overlap if near else int(overlap)}] step {step} {name}",
                            o.view(1, Hq, D).cpu().float(), q.cpu().view(1, Hq, D).contiguous(), kview, vview, n_att, 1 / math.sqrt(D), labels)
    assert more_hits > 0, "the larger resident set never produced an extra hit"
    with pytest.raises(RuntimeError):
        big.get_retrieval_position_ids(0, q)


NEAR = 128    # SKV_NEAR_SLOTS: near-miss staging slots per (batch, head) behind the E in-step slots (two lists of 64)


def _check_staging_invariant(cache, layer):
    """What the fetch launch relies on when it reads a chunk from staging instead of the host: early_of[chunk] = e implies that
    staging slot e holds that chunk's bytes and is published under its id - early_ids[e] for the in-step slots e < E (whichever
    workgroup published it: each pull workgroup of the fused selection publishes only the slots it fills, skv_early.h),
    near_pub[e - E] for the slots E .. E + 127 staged ahead by the gate/up (list 0) and down-projection (list 1) launches (round 5,
    skv_near_pull_role)."""
    ea = cache._early
    o, st = ea["offsets"], ea["states"][layer]
    B, E, nch = cache.block_num, ea["E"], ea["n_chunks"]
    ids = st[o[5]:o[5] + 4 * B * E].view(torch.int32).view(B, E).cpu()
    near = cache.near_published_ids(layer)                       # [B, 128]: list 0 | list 1
    of = st[o[6]:o[6] + 2 * B * nch].view(torch.int16).view(B, nch).cpu()
    staging = st[o[7]:o[7] + B * (E + NEAR) * 2048].view(torch.int16).view(B, E + NEAR, 1024).cpu()
    vh = cache.v_cache_cpu[layer].view(B, nch, 1024)
    for b in range(B):
        named = (of[b] >= 0).nonzero().view(-1)
        for c in named.tolist():
            e = int(of[b, c])
            assert e < E + NEAR, (b, c, e)
            pub = int(ids[b, e]) if e < E else int(near[b, e - E])
            assert pub == c, (b, c, e, pub)
            assert torch.equal(staging[b, e], vh[b, c].view(torch.int16)), (b, c, e)
        live = ids[b][ids[b] >= 0]
        assert live.unique().numel() == live.numel()             # one list: no chunk staged twice
        nlive = near[b][near[b] >= 0]
        assert nlive.unique().numel() == nlive.numel()
        for e, c in enumerate(near[b].tolist()):                 # a slot staged ahead holds its chunk whether or not the map names it
            if c >= 0:
                assert torch.equal(staging[b, E + e], vh[b, c].view(torch.int16)), (b, e, c)
        # (a chunk staged by one workgroup whose LAST step's entry another workgroup resets in the same launch ends with its
        # new entry or with -1 - the two stores are unordered across XCDs; -1 only means the fetch launch reads it from the host)
        assert set(named.tolist()) <= set(live.tolist()) | set(nlive.tolist())


def _gate_up_with_near_pull(cache, layer, g):
    """One gate/up launch (residual add + RMSNorm + [gate; up] GEMV + SiLU * mul, hidden 4096) with the near-miss pull role in
    front of its grid, against the plain launch on the same inputs: the GEMV's own result must not change."""
    from shadowkv_amd import tensor_op
    x = torch.randn(1, 1, 4096, device=DEV, generator=g).bfloat16()
    res = torch.randn(1, 1, 4096, device=DEV, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(4096, device=DEV, generator=g)).bfloat16()
    w = (torch.randn(2 * 512, 4096, device=DEV, generator=g) * 0.02).bfloat16()
    h0, y0 = tensor_op.norm_linear_decode(x, res, nw, 1e-5, w, fuse_silu_mul=True)
    args = cache.near_pull_args(layer)
    assert args is not None
    h1, y1 = tensor_op.norm_linear_decode(x, res, nw, 1e-5, w, fuse_silu_mul=True, near_pull=args)
    torch.cuda.synchronize()
    assert torch.equal(h0.view(torch.int16), h1.view(torch.int16)) and torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    # the down projection's launch (two rows per wave, the residual in the bias slot) with the pull role of list 1
    a = torch.randn(1, 1, 1024, device=DEV, generator=g).bfloat16()
    wd = (torch.randn(4096, 1024, device=DEV, generator=g) * 0.02).bfloat16()
    d0 = tensor_op.linear_decode(a, wd, bias=h0)
    args1 = cache.near_pull_args(layer, 1)
    assert args1 is not None and args1[-1] == 1
    d1 = tensor_op.linear_decode(a, wd, bias=h0, near_pull=args1)
    torch.cuda.synchronize()
    assert torch.equal(d0.view(torch.int16), d1.view(torch.int16))


@pytest.mark.parametrize("kv_heads,glm,budget", [(8, False, 2048), (4, True, 2048), (8, False, 1024)])
def test_near_miss_staging_changes_no_bit(kv_heads, glm, budget):
    """Round 5: the gate/up GEMV launch stages the chunks that fell just short of this step's selection for
    the NEXT step (skv_norm_gemv_near_pull_bf16).  Against the same steps with the early fetch alone: attention output,
    selection bookkeeping and both caches bit for bit; the GEMV's own output unchanged; the published state consistent after
    every launch; and chunks staged ahead ARE what the next step's fetch launch needed (some misses are served from them)."""
    L = 16384
    ca, cs, g = _headline_cache(kv_heads, glm, L=L, seed=41, budget=budget)
    cb, _, _ = _headline_cache(kv_heads, glm, L=L, seed=41, budget=budget)
    ca.near_lists = 2            # (both lists: the second one, staged by the down projection's launch, is off by default - it does not pay)
    ca.enable_early_fetch(near=True)
    cb.enable_early_fetch()
    assert ca.near_pull_args(0) is None and cb.near_pull_args(0) is None      # (no selection has left a near-miss list yet)
    kv_len = ca.sparse_end + 2
    q = (torch.randn(1, 32, 1, 128, device=DEV, generator=g) * 1.5).bfloat16()
    gw = torch.Generator(device=DEV).manual_seed(3)
    B, S = ca.block_num, ca.select_sets
    served = staged_total = 0
    near_quality = [[], []]
    for step in range(8):
        q = (q.float() + 0.3 * torch.randn(1, 32, 1, 128, device=DEV, generator=g)).bfloat16()
        before = [set(r[r >= 0].tolist()) for r in ca.near_published_ids(0)]        # staged ahead, as this step's fetch sees it
        oa = ca.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        ob = cb.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        torch.cuda.synchronize()
        _check_staging_invariant(ca, 0)
        assert torch.equal(oa.view(torch.int16), ob.view(torch.int16)), step
        assert torch.equal(ca.position_ids, cb.position_ids) and torch.equal(ca._cnts_layers, cb._cnts_layers)
        assert torch.equal(ca.offsets, cb.offsets)
        assert torch.equal(ca.v_cache_buffer.view(torch.int16), cb.v_cache_buffer.view(torch.int16)), step
        assert torch.equal(ca.k_cache_buffer.view(torch.int16), cb.k_cache_buffer.view(torch.int16)), step
        cnts, miss = ca.cnts.view(-1).cpu(), ca.offsets.view(B, S).cpu()
        for b in range(B):
            served += len(before[b] & set(miss[b, int(cnts[b]):].tolist()))
        # the in-step list leaves the chunks staged ahead alone
        ins = ca._early_published_ids(0)
        for b in range(B):
            assert not (set(ins[b][ins[b] >= 0].tolist()) & before[b]), (step, b)
        assert cb.near_pull_args(0) is None
        # the near-miss list this step's top-k launch left: distinct landmark chunks that were NOT selected, none scoring above
        # the S-th score, and (prediction quality, not a contract) mostly the ranks S + 1 .. S + 64 of the oracle's exact scores
        o, st = ca._early["offsets"], ca._early["states"][0]
        ncnt = st[o[10]:o[10] + 4 * 2 * B].view(torch.int32).view(2, B).cpu()
        nids = st[o[11]:o[11] + 4 * 2 * B * 64].view(torch.int32).view(2, B, 64).cpu()
        lm, lm_idx = ca.k_landmark[0][0].cpu().contiguous(), ca.k_landmark_idx[0][0].cpu()
        N, Gq = lm.shape[1], 32 // kv_heads
        Dm = torch.zeros(kv_heads, Gq, N, dtype=torch.bfloat16); P = torch.zeros_like(Dm)
        T = (N + 255) // 256
        oracle.batch_gemm_softmax(q.cpu().view(kv_heads, Gq, 128).contiguous(), lm, Dm, torch.zeros(kv_heads, T, Gq),
                                  torch.zeros(kv_heads, T, Gq), P, kv_heads, Gq, N, 128, ALPHA)
        score = P.float().max(dim=1).values                                     # [kv, N]
        sel_now = ca.position_ids[0][0].cpu()
        for b in range(B):
            slot_of = {int(c): j for j, c in enumerate(lm_idx[b].tolist())}
            srt = score[b].sort(descending=True).values
            both = []
            for k in range(2):
                n = int(ncnt[k, b])
                assert 0 <= n <= 64 and (k == 1 or n > 0), (step, b, k, n)
                ids = nids[k, b, :n].tolist()
                both += ids
                if not ids:
                    continue
                sc = torch.tensor([float(score[b, slot_of[c]]) for c in ids])
                assert float(sc.max()) <= float(srt[S - 1]), (step, b, k)
                lo = srt[min(S + 64 * (k + 1) - 1, N - 1)]
                near_quality[k].append(float((sc >= lo).float().mean()))
            assert len(set(both)) == len(both) and not (set(both) & set(sel_now[b].tolist())), (step, b)
        _gate_up_with_near_pull(ca, 0, gw)
        _check_staging_invariant(ca, 0)
        staged_total += int((ca.near_published_ids(0) >= 0).sum())
    assert staged_total > 0 and served > 0, (staged_total, served)
    for k in range(2):          # list k: mostly the ranks up to S + 64 (k + 1) of the oracle's exact scores
        assert near_quality[k] and sum(near_quality[k]) / len(near_quality[k]) > 0.8, (k, near_quality[k][:8])
    try:
        from util import open_parity_record
        with open_parity_record("near_miss_staging.txt") as f:
            f.write(f"kv {kv_heads} glm {int(glm)} budget {budget}: misses of 7 steps served from the chunks staged ahead: {served}; "
                    f"slots staged ahead after the last step: {int((ca.near_published_ids(0) >= 0).sum())} of {B * NEAR}\n")
    except OSError:
        pass


def test_near_miss_staging_survives_clear_and_a_new_prompt():
    """clear() retires the per-prompt early state with its near-miss slots; H2D() after the next prefill re-creates it (empty):
    nothing staged for the old prompt can be read for the new one."""
    ca, cs, g = _headline_cache(8, False, L=16384, seed=43)
    ca.near_lists = 2
    ca.enable_early_fetch(near=True)
    kv_len = ca.sparse_end + 2
    gw = torch.Generator(device=DEV).manual_seed(5)
    q = (torch.randn(1, 32, 1, 128, device=DEV, generator=g) * 1.5).bfloat16()
    for _ in range(3):
        q = (q.float() + 0.3 * torch.randn(1, 32, 1, 128, device=DEV, generator=g)).bfloat16()
        ca.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        _gate_up_with_near_pull(ca, 0, gw)
    assert int((ca.near_published_ids(0) >= 0).sum()) > 0
    ca.clear()
    cs, g = _headline_prefill(ca, 12288, 47)
    assert ca._early is not None and ca.near_fetch
    assert int((ca.near_published_ids(0) >= 0).sum()) == 0
    cb, _, _ = _headline_cache(8, False, L=12288, seed=47, max_length=16384)
    kv_len = ca.sparse_end + 2
    for step in range(3):
        q = (q.float() + 0.3 * torch.randn(1, 32, 1, 128, device=DEV, generator=g)).bfloat16()
        oa = ca.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        ob = cb.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        _gate_up_with_near_pull(ca, 0, gw)
        torch.cuda.synchronize()
        _check_staging_invariant(ca, 0)
        assert torch.equal(oa.view(torch.int16), ob.view(torch.int16)), step
        assert torch.equal(ca.v_cache_buffer.view(torch.int16), cb.v_cache_buffer.view(torch.int16)), step


@pytest.mark.parametrize("kv_heads,glm,budget", [(8, False, 2048), (4, True, 2048), (8, False, 4096), (4, True, 4096), (8, False, 1024)])
def test_early_fetch_changes_no_bit(kv_heads, glm, budget):
    """Speculative early V fetch (csrc/skv_early.hip): flagged in the scan launch, pulled by a launch on a side stream beside
    normalise + top-k, consumed from HBM staging by the fetch launch.  Against the same steps without it: attention output,
    selection bookkeeping and both caches bit for bit; the V rows of the selected chunks equal the host table; and the
    prediction does fire (chunks are pulled early once a previous step has left its thresholds)."""
    L = 16384 if budget <= 2048 else 32768
    ca, cs, g = _headline_cache(kv_heads, glm, L=L, seed=23, budget=budget)
    cb, _, _ = _headline_cache(kv_heads, glm, L=L, seed=23, budget=budget)
    ca.enable_early_fetch()
    kv_len = ca.sparse_end + 2
    q = (torch.randn(1, 32, 1, 128, device=DEV, generator=g) * 1.5).bfloat16()
    pulled, used = [], 0
    for step in range(8):
        if step == 5:
            q = (torch.randn(1, 32, 1, 128, device=DEV, generator=g) * 1.5).bfloat16()    # new query: the prediction is poor
        elif step == 3:
            q = q.clone()                                                                  # same query: nothing to fetch
        else:
            q = (q.float() + 0.25 * torch.randn(1, 32, 1, 128, device=DEV, generator=g)).bfloat16()
        before = ca.position_ids[0][0].clone()
        oa = ca.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        ob = cb.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        torch.cuda.synchronize()
        n_early = ca.early_fetch_counts(0)
        pulled.append(int(n_early.sum()))
        assert int(n_early.max()) <= ca._early["E"]
        assert torch.equal(ca.cnts, cb.cnts) and torch.equal(ca.offsets, cb.offsets)
        assert torch.equal(ca.position_ids, cb.position_ids)
        assert torch.equal(oa.view(torch.int16), ob.view(torch.int16)), (step, float((oa.float() - ob.float()).abs().max()))
        assert torch.equal(ca.v_cache_buffer.view(torch.int16), cb.v_cache_buffer.view(torch.int16)), step
        assert torch.equal(ca.k_cache_buffer.view(torch.int16), cb.k_cache_buffer.view(torch.int16)), step
        # every resident chunk's V rows are the host table's rows
        ids = ca.position_ids[0][0]                                                        # [kv, S] chunk ids per slot
        vh = ca.v_cache_cpu[0][0]                                                          # [kv, chunks, 8 * 128] pinned host
        for h in range(kv_heads):
            want = vh[h][ids[h].cpu()].view(-1, 128)
            got = ca.v_cache_buffer[0][0][h][ca.sparse_start:ca.sparse_start + ids.shape[1] * 8].cpu()
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (step, h)
        _check_staging_invariant(ca, 0)
        # how many of this step's misses came from staging
        new = (ca.position_ids[0][0] != before)
        used += int(new.sum())
    assert pulled[0] == 0, "no thresholds before the first step: nothing may be flagged"
    assert max(pulled[1:]) > 0, f"the prediction never fired: {pulled}"
    assert pulled[3] == 0 or pulled[3] <= pulled[2], pulled


@pytest.mark.parametrize("early_max,margin", [(1, 0.0), (128, 0.0), (16, -1.0), (16, 5.0)])
def test_early_fetch_extremes_change_no_bit(early_max, margin):
    """The early fetch at its limits: one chunk per head, the maximum of 128, a margin that flags far too much (-1: every slot
    within e of the threshold - the lists overflow and most pulled chunks are not selected) and one that flags nothing (+5).
    Outputs and caches stay those of the plain steps."""
    ca, cs, g = _headline_cache(8, False, L=16384, seed=29)
    cb, _, _ = _headline_cache(8, False, L=16384, seed=29)
    ca.enable_early_fetch(early_max=early_max, margin=margin)
    kv_len = ca.sparse_end + 2
    q = (torch.randn(1, 32, 1, 128, device=DEV, generator=g) * 1.5).bfloat16()
    pulled = 0
    for step in range(5):
        q = (q.float() + 0.3 * torch.randn(1, 32, 1, 128, device=DEV, generator=g)).bfloat16()
        oa = ca.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        ob = cb.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        torch.cuda.synchronize()
        n = ca.early_fetch_counts(0)
        assert int(n.max()) <= early_max
        pulled += int(n.sum())
        assert torch.equal(oa.view(torch.int16), ob.view(torch.int16)), step
        assert torch.equal(ca.position_ids, cb.position_ids)
        assert torch.equal(ca.v_cache_buffer.view(torch.int16), cb.v_cache_buffer.view(torch.int16)), step
        assert torch.equal(ca.k_cache_buffer.view(torch.int16), cb.k_cache_buffer.view(torch.int16)), step
    if margin >= 5.0:
        assert pulled == 0, "a threshold 5 above the k-th logit must flag nothing"
    else:
        assert pulled > 0


def test_chunk_predicted_in_two_consecutive_steps_at_different_slots_is_tolerated():
    """with the fused selection the pull workgroups of a head clear and set 16-bit early_of entries without ordering
    between them.  The case that exercises it: a chunk predicted (staged) at step t, NOT selected, and predicted again at step
    t + 1 in ANOTHER staging slot (owned by another workgroup) - its early_of entry is reset by the old owner and set by the new
    one in the same launch, and may end as the new slot or as -1.  A margin of -1 floods the lists with chunks that are not
    selected, so this happens every step; asserted: it does happen, the published state stays consistent (an entry that names
    a slot names the slot that holds the chunk; the count diagnostic equals the published slots), and selection, attention
    and both caches equal the plain steps' bit for bit."""
    ca, cs, g = _headline_cache(8, False, L=16384, seed=37)
    cb, _, _ = _headline_cache(8, False, L=16384, seed=37)
    ca.enable_early_fetch(early_max=32, margin=-1.0)
    assert ca.fused_select
    kv_len = ca.sparse_end + 2
    q = (torch.randn(1, 32, 1, 128, device=DEV, generator=g) * 1.5).bfloat16()
    prev, moved, ended_unset = None, 0, 0
    for step in range(8):
        q = (q.float() + 0.15 * torch.randn(1, 32, 1, 128, device=DEV, generator=g)).bfloat16()
        oa = ca.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        ob = cb.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        torch.cuda.synchronize()
        _check_staging_invariant(ca, 0)
        ids = ca._early_published_ids(0)                                     # [B, E]
        assert torch.equal(ca.early_fetch_counts(0), (ids >= 0).sum(dim=1).to(torch.int32))
        o, st = ca._early["offsets"], ca._early["states"][0]
        of = st[o[6]:o[6] + 2 * ca.block_num * ca._early["n_chunks"]].view(torch.int16).view(ca.block_num, -1).cpu()
        now = [{int(c): e for e, c in enumerate(row.tolist()) if c >= 0} for row in ids]
        if prev is not None:
            for b in range(ca.block_num):
                for c, e in now[b].items():
                    if c in prev[b] and prev[b][c] != e:
                        moved += 1
                        assert int(of[b, c]) in (-1, e), (step, b, c, e, int(of[b, c]))
                        ended_unset += int(of[b, c]) == -1
        prev = now
        assert torch.equal(oa.view(torch.int16), ob.view(torch.int16)), step
        assert torch.equal(ca.position_ids, cb.position_ids) and torch.equal(ca._cnts_layers, cb._cnts_layers)
        assert torch.equal(ca.v_cache_buffer.view(torch.int16), cb.v_cache_buffer.view(torch.int16)), step
        assert torch.equal(ca.k_cache_buffer.view(torch.int16), cb.k_cache_buffer.view(torch.int16)), step
    assert moved > 0, "no chunk was predicted twice in a row at different slots: the scenario was not exercised"
    try:
        from util import open_parity_record
        with open_parity_record("early_fetch_moved_entries.txt") as f:
            f.write(f"chunks predicted in consecutive steps at different staging slots: {moved}; entry ended as -1: {ended_unset}\n")
    except OSError:
        pass


@pytest.mark.parametrize("kv_heads,glm", [(8, False), (4, True)])
def test_early_fetch_with_thresholds_that_jump_changes_no_selection(kv_heads, glm):
    """Query scales that move the k-th score far up and far down between steps, so that the early fetch's thresholds (taken
    from the previous step) flag nearly everything or nothing: selection, bookkeeping and caches equal the plain steps'."""
    ca, cs, g = _headline_cache(kv_heads, glm, L=16384, seed=31)
    cb, _, _ = _headline_cache(kv_heads, glm, L=16384, seed=31)
    ca.enable_early_fetch(early_max=4)
    kv_len = ca.sparse_end + 2
    base = torch.randn(1, 32, 1, 128, device=DEV, generator=g)
    for step, scale in enumerate([1.5, 1.5, 0.2, 0.2, 3.0, 0.05, 1.0, 4.0, 4.0, 0.5]):
        q = ((base + 0.2 * torch.randn(1, 32, 1, 128, device=DEV, generator=g)) * scale).bfloat16()
        oa = ca.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        ob = cb.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        torch.cuda.synchronize()
        assert torch.equal(ca.cnts, cb.cnts), (step, scale)
        assert torch.equal(ca.offsets, cb.offsets), (step, scale)
        assert torch.equal(ca.position_ids, cb.position_ids), (step, scale)
        assert torch.equal(oa.view(torch.int16), ob.view(torch.int16)), (step, scale)
    assert torch.equal(ca.v_cache_buffer.view(torch.int16), cb.v_cache_buffer.view(torch.int16))


@pytest.mark.parametrize("first,second", [(12288, 16384), (16384, 10240)])
def test_early_fetch_survives_clear_and_a_prompt_of_another_length(first, second):
    """clear() + a new prefill with the early fetch enabled (the state is carved for one prompt's landmark and chunk
    counts; the reference's evaluation loop clears and re-prefills per sample, test/evaluator.py:83).  clear() retires the state,
    H2D() after the next prefill re-creates it for the new landmark count - also when that count grows past a 256-slot tile
    boundary -, and the steps of the second prompt equal, bit for bit, those of a cache that never had the early fetch; a state
    that does not match the cache is refused before any launch."""
    ca, cs, g = _headline_cache(8, False, L=first, seed=41, max_length=16384)
    cb, _, _ = _headline_cache(8, False, L=first, seed=41, max_length=16384)
    ca.enable_early_fetch(early_max=24)
    n_lm_first = ca._early["n_lm"]
    kv_len = ca.sparse_end + 2

    def steps(n, q):
        pulled = 0
        for step in range(n):
            q = (q.float() + 0.25 * torch.randn(1, 32, 1, 128, device=DEV, generator=g)).bfloat16()
            oa = ca.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
            ob = cb.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
            torch.cuda.synchronize()
            pulled += int(ca.early_fetch_counts(0).sum())
            assert torch.equal(ca.cnts, cb.cnts) and torch.equal(ca.offsets, cb.offsets), step
            assert torch.equal(ca.position_ids, cb.position_ids), step
            assert torch.equal(oa.view(torch.int16), ob.view(torch.int16)), step
            assert torch.equal(ca.v_cache_buffer.view(torch.int16), cb.v_cache_buffer.view(torch.int16)), step
            assert torch.equal(ca.k_cache_buffer.view(torch.int16), cb.k_cache_buffer.view(torch.int16)), step
        return pulled

    q0 = (torch.randn(1, 32, 1, 128, device=DEV, generator=g) * 1.5).bfloat16()
    assert steps(3, q0) > 0
    ca.clear(); cb.clear()
    assert ca._early is None and ca._early_request == (24, 0.0) and ca._pending_v is None and ca._pushed is None
    _headline_prefill(ca, second, seed=43)
    _, g = _headline_prefill(cb, second, seed=43)
    assert ca._early is not None and ca._early["E"] == 24 and ca._early["n_lm"] == ca.k_landmark.shape[-2] != n_lm_first
    assert steps(4, q0) > 0, "the prediction never fired on the second prompt"
    # a state built for another landmark count is refused (a new prefill WITHOUT clear() would leave one behind)
    ca._early["n_lm"] = n_lm_first
    with pytest.raises(RuntimeError, match="early-fetch state was built for"):
        ca.select_fetch_attend_inplace(0, q0, cs, kv_len=kv_len)
    ca._early["n_lm"] = ca.k_landmark.shape[-2]
    ca.enable_early_fetch(0)
    assert ca._early is None and ca._early_request is None


def test_early_fetch_landmark_map_reproduces_the_slot_to_chunk_ids():
    """skv_early_state_set_landmark_map (round 4): the list role of the early fetch computes a flagged slot's chunk id as
    slot + #{gaps <= slot} from a table of the chunks the landmark sequence leaves out (the outliers) instead of gathering
    k_landmark_idx.  The table must reproduce k_landmark_idx exactly for every slot of every head; a landmark_idx that is not
    "ascending chunk ids minus a few" is flagged on the device and keeps the gather."""
    from shadowkv_amd import _lib
    for kv_heads, glm, budget, L in ((8, False, 2048, 16384), (4, True, 4096, 32768)):
        cache, cs, g = _headline_cache(kv_heads, glm, L=L, seed=7, budget=budget)
        cache.enable_early_fetch()
        torch.cuda.synchronize()
        st, o = cache._early["states"][0], cache._early["offsets"]
        B, N = cache.block_num, cache.k_landmark.shape[-2]
        assert len(o) == 13
        gaps = st[o[8]:o[8] + 4 * B * 128].view(torch.int32).view(B, 128).cpu()
        ok = st[o[9]:o[9] + 4 * B].view(torch.int32).cpu()
        assert ok.tolist() == [1] * B
        idx = cache.k_landmark_idx[0][0].cpu()                                    # [kv, N]
        slot = torch.arange(N)
        for h in range(B):
            assert bool((gaps[h][:-1] <= gaps[h][1:]).all())
            n_gaps = int((gaps[h] < 2 ** 31 - 1).sum())
            assert 0 < n_gaps <= budget // 1024 * 24          # the outlier chunks (kv_cache.py:548) below the last landmark
            ids = slot + (gaps[h].unsqueeze(0) <= slot.unsqueeze(1)).sum(dim=1)
            assert torch.equal(ids, idx[h]), f"head {h}: the gap table does not reproduce k_landmark_idx"
    # a landmark_idx that is not ascending: detected, the map is off (the list role gathers as before)
    L_ = _lib.lib()
    B, G, N, n_chunks, E = 2, 4, 1000, 1100, 8
    state = torch.empty(L_.skv_early_state_bytes(B, G, N, n_chunks, E), dtype=torch.uint8, device=DEV)
    _lib.check(L_.skv_early_state_init(state.data_ptr(), B, G, N, n_chunks, E, 0), "early_state_init")
    offs = (ctypes.c_longlong * 10)()
    _lib.check(L_.skv_early_state_offsets2(B, G, N, n_chunks, E, offs, 10), "early_state_offsets2")
    off8 = (ctypes.c_longlong * 9)(*([-7] * 9))                 # the round-3 name keeps its eight-entry contract
    _lib.check(L_.skv_early_state_offsets(B, G, N, n_chunks, E, ctypes.cast(off8, ctypes.POINTER(ctypes.c_longlong))), "early_state_offsets")
    assert list(off8)[:8] == list(offs)[:8] and off8[8] == -7
    good = torch.arange(N, dtype=torch.int64) + (torch.arange(N) >= 500) * 3
    bad = good.clone(); bad[10], bad[11] = good[11], good[10]
    lm = torch.stack([good, bad]).to(DEV)
    _lib.check(L_.skv_early_state_set_landmark_map(state.data_ptr(), lm.data_ptr(), B, G, N, n_chunks, E, 0), "set_landmark_map")
    torch.cuda.synchronize()
    assert state[offs[9]:offs[9] + 8].view(torch.int32).cpu().tolist() == [1, 0]
    assert state[offs[8]:offs[8] + 12].view(torch.int32).cpu().tolist() == [500, 500, 500]
    # ids BELOW their slot index (d_j - of the previous slot - negative while d_j itself is in range) on head 0 AND
    # head 1: flagged, and no byte in front of / outside a head's own gap table is written (head 0's table is preceded by
    # the staging, early_of and early_ids regions; head 1's by head 0's table)
    for kind in ("zeros", "duplicates", "one_low"):
        _lib.check(L_.skv_early_state_init(state.data_ptr(), B, G, N, n_chunks, E, 0), "early_state_init")
        low = good.clone()
        if kind == "zeros":
            low[200:260] = 0                                       # d = -j over a run, then back in range at slot 260
        elif kind == "duplicates":
            low[1:] = good[:-1].clone(); low[300:] = good[300]     # one id repeated to the end: d falls below zero
        else:
            low[700] = 5                                           # a single id far below its slot; slot 701 has dp = -695, d = 3
        lm = torch.stack([low, low]).to(DEV)
        torch.cuda.synchronize()
        before = state.clone()
        _lib.check(L_.skv_early_state_set_landmark_map(state.data_ptr(), lm.data_ptr(), B, G, N, n_chunks, E, 0), "set_landmark_map")
        torch.cuda.synchronize()
        assert state[offs[9]:offs[9] + 8].view(torch.int32).cpu().tolist() == [0, 0], kind
        lo, hi = offs[8], offs[8] + 4 * B * 128                   # the gap tables
        assert torch.equal(state[:lo], before[:lo]), f"{kind}: bytes in front of the gap tables changed"
        rest = torch.ones(state.numel() - hi, dtype=torch.bool, device=DEV)
        rest[offs[9] - hi:offs[9] - hi + 4 * B] = False            # (map_ok itself is written)
        assert torch.equal(state[hi:][rest], before[hi:][rest]), f"{kind}: bytes behind the gap tables changed"
