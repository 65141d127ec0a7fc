"""Times ShadowKVCache_CPU.prefill_kv_cache's chunk statistics at the headline context (one layer, 8 kv heads x
124,928 keys): the native pass (skv_chunk_stats) against the chain of ATen ops it replaces (kv_cache.py:854-868)."""
import torch
from shadowkv_amd import tensor_op

dev = "cuda:0"
L, kv, C, D = 124928, 8, 8, 128
chunks = L // C - 4
chunks -= chunks % 8
k = torch.randn(1, kv, L, D, device=dev).bfloat16()
kc = k[:, :, : chunks * C]


def aten():
    kk = kc.view(1, kv, chunks, C, D)
    m = kk.mean(dim=-2)
    return m, torch.nn.functional.cosine_similarity(m.unsqueeze(3).expand(-1, -1, -1, C, -1), kk, dim=-1).min(-1).values


def native():
    return tensor_op.chunk_stats(kc, C)


for name, fn in (("aten ops", aten), ("skv_chunk_stats", native)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:<16} {ms*1e3:9.1f} us per layer   ({kc.numel()*2/ms/1e9:.2f} TB/s of K)", flush=True)
