"""Where does the host time of the drop-in path go?  cProfile of DecoderLM.decode_step(fused=False) (the reference's call
order, eager) at the headline configuration; prints ms/step and the functions by own time.
usage (GPU box, repo root): python tools/call_order_profile.py [layers]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from shadowkv_amd import llama  # noqa: E402

dev = "cuda:0"
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = 122 * 1024
m = llama.DecoderLM(cfg=llama.LLAMA_3_1_8B, batch_size=1, max_length=ctx, device=dev, sparse_budget=2048, rank=160,
                    chunk_size=8, num_layers=layers, chunk_layout="inplace", overlap_attention=True)
llama.build_synthetic_context(m, ctx, seed=4321)
walk = llama.QueryWalk(m, step=0.3, seed=99)
m.query_hook = walk
tok = torch.randint(0, m.cfg.vocab_size, (1, 1), device=dev)


def step():
    global tok
    walk.advance()
    tok = m.decode_step(tok, temperature=0.6, fused=False)
    return tok[:, -1].tolist()


for _ in range(6):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(16):
    step()
torch.cuda.synchronize()
print(f"call order, eager: {(time.perf_counter() - t0) / 16 * 1e3:.3f} ms/step ({layers} layers)")
# host time only: how long the Python side takes to ISSUE a step (no synchronisation inside)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    walk.advance()
    tok = m.decode_step(tok, temperature=0.6, fused=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host issue time: {(t1 - t0) / 8 * 1e3:.3f} ms/step; with the final sync {(time.perf_counter() - t0) / 8 * 1e3:.3f}")
m.kv_cache.gen_offset = 0
m.kv_cache.kv_offset = ctx
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
