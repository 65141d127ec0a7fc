"""GPU: the hipGraph-captured decode step (GraphDecoder) must reproduce the eager step bit for bit:
same tokens, same chunk bookkeeping, same cache bytes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(seed=5):
    from shadowkv_amd import llama
    cfg = llama.ModelConfig(name="tiny", hidden_size=1024, intermediate_size=2048, num_hidden_layers=2,
                            num_attention_heads=8, num_key_value_heads=2, vocab_size=2000)
    m = llama.DecoderLM(cfg=cfg, batch_size=1, max_length=4608, device=DEV, sparse_budget=256, rank=160, chunk_size=8,
                        seed=seed)
    llama.build_synthetic_context(m, 4608, seed=77)
    return m, llama


@pytest.mark.parametrize("use_walk", [False, True])
def test_graph_equals_eager(use_walk):
    steps = 6
    m1, llama = _make()
    table = llama.make_walk_table(m1, steps, seed=3) if use_walk else None
    tok0 = torch.tensor([[17]], device=DEV)
    # eager reference
    toks1 = []
    t = tok0.clone()
    for i in range(steps):
        if use_walk:
            m1.query_hook = (lambda tbl, i: (lambda l, q: torch.addcmul(tbl[i][l], q, torch.zeros((), device=DEV, dtype=q.dtype))))(table, i)
        t = m1.decode_step(t, temperature=0.0)
        toks1.append(int(t))
    torch.cuda.synchronize()
    # graph
    m2, _ = _make()
    dec = llama.GraphDecoder(m2, temperature=0.0, walk_table=table)
    dec.token.copy_(tok0)
    warm = dec.capture(warmup=2)
    toks2 = []
    # the two warm-up steps already produced tokens; re-run from scratch is not possible, so compare the tail
    for _ in range(steps - warm):
        toks2.append(int(dec.step()))
    torch.cuda.synchronize()
    assert toks2 == toks1[warm:], (toks1, toks2)
    c1, c2 = m1.kv_cache, m2.kv_cache
    assert c1.kv_offset == c2.kv_offset and c1.gen_offset == c2.gen_offset
    assert torch.equal(c1.position_ids, c2.position_ids)
    assert torch.equal(c1.k_cache_buffer.view(torch.int16), c2.k_cache_buffer.view(torch.int16))
    assert torch.equal(c1.v_cache_buffer.view(torch.int16), c2.v_cache_buffer.view(torch.int16))
