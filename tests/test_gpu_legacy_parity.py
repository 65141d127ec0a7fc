"""GPU parity: every `kernels.shadowkv` entry point (C ABI part 1) against the CPU oracle.

Integer / byte / index outputs and everything on the selection path must be bit-exact; the
MFMA K rebuild is compared in bf16 ulps (tolerance written at the assert)."""
import math

import pytest
import torch

import oracle
from util import ALPHA, assert_bits_equal, ulp_diff_bf16, make_selection_step, open_parity_record, REBUILD_FLIP_BOUND

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def sk():
    from shadowkv_amd.kernels import shadowkv
    return shadowkv


@pytest.mark.parametrize("B,m,n", [(8, 4, 1500), (2, 4, 256), (3, 8, 777), (1, 1, 300), (2, 16, 513), (8, 4, 15560)])
def test_batch_gemm_softmax(sk, B, m, n):
    g = torch.Generator().manual_seed(n + m)
    q = (torch.randn(B, m, 128, generator=g) * 2).bfloat16()
    lm = torch.randn(B, n, 128, generator=g).bfloat16()
    T = (n + 255) // 256
    D0 = torch.zeros(B, m, n, dtype=torch.bfloat16); P0 = torch.zeros_like(D0)
    N0 = torch.zeros(B, T, m); S0 = torch.zeros(B, T, m)
    oracle.batch_gemm_softmax(q, lm, D0, N0, S0, P0, B, m, n, 128, ALPHA)
    D1 = torch.zeros(B, m, n, dtype=torch.bfloat16, device=DEV); P1 = torch.zeros_like(D1)
    N1 = torch.zeros(B, T, m, device=DEV); S1 = torch.zeros(B, T, m, device=DEV)
    sk.batch_gemm_softmax(q.to(DEV), lm.to(DEV), D1, N1, S1, P1, B, m, n, 128, ALPHA, 0)
    torch.cuda.synchronize()
    assert_bits_equal(D0, D1, "logits D")
    assert torch.equal(N0, N1.cpu()), "Norm (partials + final max) differ"
    assert torch.equal(S0, S1.cpu()), "Sum (partials + final 1/sum) differ"
    assert_bits_equal(P0, P1, "softmax P")


@pytest.mark.parametrize("S", [32, 128, 256, 100, 512, 1024])
@pytest.mark.parametrize("hit", [0.0, 0.6, 1.0])
def test_reorder(sk, S, hit):
    g = torch.Generator().manual_seed(S * 7 + int(hit * 10))
    B = 16
    cached, cur = [], []
    for _ in range(B):
        c, n = make_selection_step(g, 16000, S, hit)
        cached.append(c); cur.append(n)
    cached = torch.stack(cached); cur = torch.stack(cur)
    c0 = cached.clone(); off0 = torch.zeros(B, S, dtype=torch.int32); cnt0 = torch.zeros(B, dtype=torch.int32)
    oracle.reorder_keys_and_compute_offsets(c0, cur, off0, cnt0, 2, 8, S)
    c1 = cached.to(DEV); off1 = torch.zeros(B * S, dtype=torch.int32, device=DEV)
    cnt1 = torch.zeros(B, dtype=torch.int32, device=DEV)
    sk.reorder_keys_and_compute_offsets(c1, cur.to(DEV), off1, cnt1, 2, 8, S)
    torch.cuda.synchronize()
    assert torch.equal(cnt0, cnt1.cpu())
    assert torch.equal(c0, c1.cpu())
    assert torch.equal(off0.flatten(), off1.cpu())


def test_reorder_initial_minus_one(sk):
    """position_ids start as -1 (kv_cache.py:634): nothing matches, all misses, sorted."""
    S, B = 256, 8
    cached = torch.full((B, S), -1, dtype=torch.int64)
    g = torch.Generator().manual_seed(5)
    cur = torch.stack([torch.randperm(15000, generator=g)[:S] for _ in range(B)]).to(torch.int64)
    c1 = cached.to(DEV); off = torch.zeros(B * S, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(B, dtype=torch.int32, device=DEV)
    sk.reorder_keys_and_compute_offsets(c1, cur.to(DEV), off, cnt, 1, B, S)
    torch.cuda.synchronize()
    assert int(cnt.sum()) == 0
    assert torch.equal(c1.cpu(), cur.sort(dim=-1).values)
    assert torch.equal(off.cpu().view(B, S).to(torch.int64), cur.sort(dim=-1).values)


def _mover_case(g, B, S, hit, n_host_chunks, buf_rows, off_rows):
    D = 128
    host = torch.randint(-30000, 30000, (B, n_host_chunks, 8 * D), generator=g, dtype=torch.int16).view(torch.bfloat16)
    buf = torch.randint(-30000, 30000, (B, buf_rows, D), generator=g, dtype=torch.int16).view(torch.bfloat16)
    cached, cur = [], []
    for _ in range(B):
        c, n = make_selection_step(g, n_host_chunks, S, hit)
        cached.append(c); cur.append(n)
    cached = torch.stack(cached); cur = torch.stack(cur)
    off = torch.zeros(B, S, dtype=torch.int32); cnt = torch.zeros(B, dtype=torch.int32)
    oracle.reorder_keys_and_compute_offsets(cached, cur, off, cnt, 1, B, S)
    return host, buf, off, cnt


@pytest.mark.parametrize("S,hit", [(256, 0.6), (256, 0.0), (256, 1.0), (32, 0.5), (100, 0.3)])
def test_gather_copy_with_offsets(sk, S, hit):
    g = torch.Generator().manual_seed(S + int(hit * 100))
    B, nch, D = 8, 2000, 128
    off_rows = 72
    buf_rows = off_rows + S * 8 + 40
    host, buf, off, cnt = _mover_case(g, B, S, hit, nch, buf_rows, off_rows)
    b0 = buf.clone()
    args = (1, B, nch * 8 * D, S * 8 * D, off_rows * D, buf_rows * D, S)
    oracle.gather_copy_with_offsets(host, b0, None, off, cnt, None, *args)
    hostp = host.pin_memory()
    b1 = buf.to(DEV); sig = torch.zeros(B, dtype=torch.int32, device=DEV)
    temp = torch.zeros(B, S, 8 * D, dtype=torch.bfloat16, device=DEV)     # the reference's bounce buffer shape
    sk.gather_copy_with_offsets(hostp, b1, temp, off.to(DEV), cnt.to(DEV), sig, *args)
    torch.cuda.synchronize()
    assert_bits_equal(b0, b1, "V buffer after gather_copy_with_offsets")
    assert int(sig.abs().sum()) == 0, "signals must be zero on exit"
    with pytest.raises(ValueError):                                        # an undersized bounce buffer is refused
        sk.gather_copy_with_offsets(hostp, b1, temp[:1, :1], off.to(DEV), cnt.to(DEV), sig, *args)


@pytest.mark.parametrize("S,hit", [(256, 0.6), (256, 1.0), (128, 0.2)])
def test_gather_copy_d2d(sk, S, hit):
    g = torch.Generator().manual_seed(S * 3 + int(hit * 100))
    B, nch, D = 8, 2000, 128
    off_rows = 448
    buf_rows = off_rows + S * 8 + 96
    _, buf, off, cnt = _mover_case(g, B, S, hit, nch, buf_rows, off_rows)
    b0 = buf.clone()
    args = (1, B, S * 8 * D, off_rows * D, buf_rows * D, S)
    oracle.gather_copy_d2d_with_offsets(b0, off, cnt, *args)
    b1 = buf.to(DEV)
    sk.gather_copy_d2d_with_offsets(b1, off.to(DEV), cnt.to(DEV), *args)
    torch.cuda.synchronize()
    assert_bits_equal(b0, b1, "K buffer after gather_copy_d2d_with_offsets")


def test_gather_copy_nocache(sk):
    g = torch.Generator().manual_seed(11)
    B, nch, S, D = 8, 3000, 256, 128
    host = torch.randint(-30000, 30000, (B, nch, 8 * D), generator=g, dtype=torch.int16).view(torch.bfloat16)
    ids = torch.stack([torch.randperm(nch, generator=g)[:S] for _ in range(B)]).to(torch.int64)
    out0 = torch.zeros(B, S * 8, D, dtype=torch.bfloat16)
    oracle.gather_copy(host, out0, ids, 1, B, nch * 8 * D, S * 8 * D, S)
    out1 = torch.zeros(B, S * 8, D, dtype=torch.bfloat16, device=DEV)
    sk.gather_copy(host.pin_memory(), out1, ids.to(DEV), 1, B, nch * 8 * D, S * 8 * D, S)
    torch.cuda.synchronize()
    assert_bits_equal(out0, out1, "gather_copy")


def _rebuild_inputs(g, bs, heads, L, R, S, C):
    U = (torch.randn(bs, L, R, generator=g) / math.sqrt(R)).bfloat16()
    SV = torch.randn(bs, heads, 128, R, generator=g).bfloat16()
    ids = torch.stack([torch.randperm(L // C, generator=g)[:S] for _ in range(bs * heads)]).view(bs, heads, S)
    cnts = torch.randint(0, S + 1, (bs * heads,), generator=g, dtype=torch.int32)
    cnts[0] = 0
    cnts[-1] = S
    return U, SV, ids.to(torch.int32), cnts


@pytest.mark.parametrize("bs,heads,L,S", [(1, 8, 4096, 256), (2, 4, 2048, 32), (1, 2, 1024, 100)])
def test_batch_gather_gemm(sk, bs, heads, L, S):
    g = torch.Generator().manual_seed(L + S)
    R, C = 160, 8
    U, SV, ids, cnts = _rebuild_inputs(g, bs, heads, L, R, S, C)
    out0 = torch.zeros(bs, heads, S * C, 128, dtype=torch.bfloat16)
    oracle.batch_gather_gemm(U, SV, None, None, ids, out0, bs, heads, L, 128, R, S * C, L, C, cnts)
    out1 = torch.zeros(bs, heads, S * C, 128, dtype=torch.bfloat16, device=DEV)
    cs = torch.zeros(1, dtype=torch.bfloat16, device=DEV)
    sk.batch_gather_gemm(U.to(DEV), SV.to(DEV), cs, cs, ids.to(DEV), out1, bs, heads, L, 128, R, S * C, L, C,
                         cnts.to(DEV))
    torch.cuda.synchronize()
    out1 = out1.cpu()
    # only rows of chunks >= cnt are defined
    tot, bad1 = 0, 0
    worst = 0
    for b in range(bs):
        for h in range(heads):
            r0 = int(cnts[b * heads + h]) * C
            # untouched hit rows must stay zero
            assert int(out1[b, h, :r0].abs().sum()) == 0
            if r0 == S * C:
                continue
            pos = (ids[b, h].to(torch.int64)[:, None] * C + torch.arange(C)[None, :]).flatten()[r0:]
            sabs = U[b, pos].float().abs() @ SV[b, h].float().abs().T   # sum_j |u_j * sv_j| per output
            d = ulp_diff_bf16(out0[b, h, r0:], out1[b, h, r0:])
            adiff = (out0[b, h, r0:].float() - out1[b, h, r0:].float()).abs()
            # f32 accumulation order differs (hardware MFMA order vs the oracle's model of it): the
            # two f32 sums differ by at most a few f32 ulps of sum|products| (2^-20 * sabs is generous),
            # which can flip one bf16 rounding (1 ulp) or, near zero, show up as that absolute error.
            ok = (d <= 1) | (adiff <= 2.0 ** -20 * sabs)
            assert bool(ok.all()), f"{int((~ok).sum())} values outside tolerance"
            tot += d.numel(); bad1 += int((d >= 1).sum())
            # pre-RoPE output is directly comparable: away from zero (|x| >= 2^-10 sum|products|) nothing may be more than
            # ONE bf16 ulp off (a lost k-step or lane quarter would be)
            big = out0[b, h, r0:].float().abs() >= 2.0 ** -10 * sabs
            worst = max(worst, int(d[big].max()) if bool(big.any()) else 0)
    with open_parity_record() as f:
        f.write(f"{'test_batch_gather_gemm[%d-%d-%d-%d]' % (bs, heads, L, S):72s} {'pre-RoPE':14s} values {tot:9d}  "
                f"differing {bad1 / max(tot, 1):.6f}  max ulp {worst}\n")
    assert worst <= 1, f"a pre-RoPE value is {worst} bf16 ulps off"
    assert bad1 <= REBUILD_FLIP_BOUND * tot, f"{bad1}/{tot} values differ by 1 bf16 ulp"


def _cos_sin(L, width, g):
    return torch.randn(L, width, generator=g).clamp(-1, 1).bfloat16()


@pytest.mark.parametrize("glm", [False, True])
def test_rope_push_cache(sk, glm):
    g = torch.Generator().manual_seed(3 + glm)
    bs, heads, S, C, L = 2, 4, 32, 8, 4096
    x = torch.randn(bs, heads, S * C, 128, generator=g).bfloat16()
    cs = _cos_sin(L, 64 if glm else 128, g)
    ids = torch.stack([torch.randperm(L // C, generator=g)[:S] for _ in range(bs * heads)]).view(bs, heads, S).to(torch.int32)
    cnts = torch.randint(0, S, (bs * heads,), generator=g, dtype=torch.int32)
    buf_rows, start = 64 + S * C + 16, 64
    cache0 = torch.randn(bs, heads, buf_rows, 128, generator=g).bfloat16()
    cache1 = cache0.to(DEV)
    ints = (bs, heads, S * C, 128, x.stride(0), x.stride(1), x.stride(2), x.stride(3), cs.stride(0),
            ids.stride(0), ids.stride(1), ids.stride(2), cache0.stride(0), cache0.stride(1), cache0.stride(2),
            start, start + S * C, 64, C)
    fn0 = oracle.apply_rotary_pos_emb_push_cache_opt_glm if glm else oracle.apply_rotary_pos_emb_push_cache_opt
    fn1 = sk.apply_rotary_pos_emb_push_cache_opt_glm if glm else sk.apply_rotary_pos_emb_push_cache_opt
    fn0(x, cs, ids, cache0, cnts, *ints)
    fn1(x.to(DEV), cs.to(DEV), ids.to(DEV), cache1, cnts.to(DEV), *ints)
    torch.cuda.synchronize()
    assert_bits_equal(cache0, cache1, "key cache after RoPE push")


def test_rope_new_and_v1(sk):
    g = torch.Generator().manual_seed(9)
    bs, heads, s, L = 2, 8, 40, 5000
    x = torch.randn(bs, heads, s, 128, generator=g).bfloat16()
    cs = _cos_sin(L, 128, g)
    pid = torch.randint(0, L, (bs, heads, s), generator=g, dtype=torch.int64)
    o0 = torch.zeros_like(x)
    ints = (bs, heads, s, 128, x.stride(0), x.stride(1), x.stride(2), x.stride(3), cs.stride(0), pid.stride(0),
            pid.stride(1), pid.stride(2), 64)
    oracle.apply_rotary_pos_emb_new(x, cs, pid, o0, *ints)
    o1 = torch.zeros_like(x, device=DEV)
    sk.apply_rotary_pos_emb_new(x.to(DEV), cs.to(DEV), pid.to(DEV), o1, *ints)
    # separate full-width tables built from the fused one must give the same result
    cosf = torch.cat((cs[:, :64], cs[:, :64]), -1).contiguous(); sinf = torch.cat((cs[:, 64:], cs[:, 64:]), -1).contiguous()
    o2 = torch.zeros_like(x, device=DEV)
    sk.apply_rotary_pos_emb(x.to(DEV), cosf.to(DEV), sinf.to(DEV), pid.to(DEV), o2, bs, heads, s, 128, x.stride(0),
                            x.stride(1), x.stride(2), x.stride(3), 128, 128, pid.stride(0), pid.stride(1),
                            pid.stride(2), 64)
    torch.cuda.synchronize()
    assert_bits_equal(o0, o1, "apply_rotary_pos_emb_new")
    assert_bits_equal(o0, o2, "apply_rotary_pos_emb (separate tables)")


def test_unsupported_shapes_raise(sk):
    from shadowkv_amd._lib import ShadowKVNativeError
    z = torch.zeros(8, dtype=torch.int64, device=DEV)
    zi = torch.zeros(8, dtype=torch.int32, device=DEV)
    with pytest.raises(ShadowKVNativeError):
        sk.reorder_keys_and_compute_offsets(z, z, zi, zi, 1, 1, 2048)  # map_size > 1024
    a = torch.zeros(1, 4, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ShadowKVNativeError):
        sk.batch_gemm_softmax(a, a, a, torch.zeros(4, device=DEV), torch.zeros(4, device=DEV), a, 1, 4, 1, 64, 1.0, 0)
    with pytest.raises(TypeError):
        sk.reorder_keys_and_compute_offsets(zi, z, zi, zi, 1, 1, 8)  # wrong dtype
