"""GPU, BASELINE.json sizes (Llama-3.1-8B shapes, 124,928-token context, budget 2048, rank 160; 2 layers to keep the
run short): size-independent properties of the decode path after several captured steps - the oracle cannot follow at
this size in seconds, so the checks are invariants the domain offers:
  * every slot of the sparse region holds exactly the V chunk its position_ids entry names (bytes from the host table)
  * its K rows equal RoPE(U[rows] . SV^T) recomputed with PyTorch for the same ids (tolerance of the MFMA order)
  * position_ids of a head are distinct, in range, and none is an outlier chunk (all are landmark ids)
  * the K rebuild is homogeneous: scaling SV by 2 scales every rebuilt key by exactly 2 (bf16 scaling is exact and
    commutes with every rounding point), bit for bit, at full U size (64-bit row offsets)
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CTX = 124928


@pytest.fixture(scope="module")
def decoded():
    from shadowkv_amd import llama
    m = llama.DecoderLM(cfg=llama.LLAMA_3_1_8B, batch_size=1, max_length=CTX, device=DEV, sparse_budget=2048, rank=160,
                        chunk_size=8, num_layers=2, seed=3, chunk_layout="inplace", overlap_attention=True)
    llama.build_synthetic_context(m, CTX, seed=11)
    table = llama.make_walk_table(m, 12, seed=5)
    dec = llama.GraphDecoder(m, temperature=0.6, walk_table=table)
    dec.token.copy_(torch.tensor([[7]], device=DEV))
    dec.capture()
    for _ in range(6):
        dec.step()
    torch.cuda.synchronize()
    return m


def test_sparse_region_holds_the_chunks_its_ids_name(decoded):
    m = decoded
    c = m.kv_cache
    C, D, S = c.chunk_size, c.head_dim, c.select_sets
    assert (c.sparse_start, c.sparse_end, S) == (448, 2496, 256)
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
            cos, sin = cs[:, :64], cs[:, 64:]
            x1, x2 = k_pre[:, :64].float(), k_pre[:, 64:].float()
            want_k = torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1)
            got_k = c.k_cache_buffer[l][0, h, c.sparse_start:c.sparse_end].float()
            bound = 2.0 ** -6 * torch.cat((x1.abs() + x2.abs(),) * 2, dim=-1) + 1e-3
            assert bool(((got_k - want_k).abs() <= bound).all()), (l, h, float(((got_k - want_k).abs() - bound).max()))


def test_generated_rows_and_counters(decoded):
    m = decoded
    c = m.kv_cache
    n = 8                                                                  # 2 eager warm-up steps + 6 replays
    assert c.kv_offset == CTX + n and c.gen_offset == n
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
