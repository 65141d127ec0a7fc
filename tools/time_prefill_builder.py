"""Times ShadowKVCache_CPU.prefill_kv_cache (one layer, headline shape) and its parts with CUDA events."""
import math, sys, time, torch
sys.path.insert(0, ".")
from shadowkv_amd import llama, tensor_op
dev = "cuda:0"
L = 124928
m = llama.DecoderLM(cfg=llama.LLAMA_3_1_8B, batch_size=1, max_length=L, device=dev, num_layers=1, seed=1)
c = m.kv_cache
kv, D, r = 8, 128, 160
g = torch.Generator(device=dev).manual_seed(0)
k = torch.randn(1, kv, L, D, device=dev, generator=g).bfloat16()
v = torch.randn(1, kv, L, D, device=dev, generator=g).bfloat16()
q = torch.randn(1, 32, 1, D, device=dev, generator=g).bfloat16()
c.U = torch.zeros(1, 1, L, r, device=dev, dtype=torch.bfloat16); c.SV = torch.zeros(1, 1, kv, D, r, device=dev, dtype=torch.bfloat16)
def run():
    c.prefilled_batch = 0
    c.prefill_kv_cache(v, 0, k, q)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
print(f"prefill_kv_cache: {(time.perf_counter()-t0)/5*1e3:.2f} ms per layer (V to pinned host: {v.numel()*2/1e6:.0f} MB)")
t0 = time.perf_counter()
for _ in range(5):
    c.v_cache_cpu[0][:, :, :L // 8].copy_(v.reshape(1, kv, L // 8, 8 * D), non_blocking=True)
torch.cuda.synchronize()
print(f"  V chunk table D2H copy alone: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
