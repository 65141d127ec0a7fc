#!/usr/bin/env python3
"""In-kernel phase stamps of the fused fetch launch (K rebuild || V fetch || attention) at the headline shape.
Build the diagnostic library first: make -C shadowkv_amd/csrc stamps ; run with SKV_LIB_PATH=shadowkv_amd/libshadowkv_hip_stamps.so"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
from shadowkv_amd import _lib
from test_gpu_kv_cache import _headline_cache

def main():
    hit = float(sys.argv[1]) if len(sys.argv) > 1 else 0.67
    kv = int(sys.argv[2]) if len(sys.argv) > 2 else 8            # 8: Llama (G = 4); 4: GLM-4 (G = 8, GLM RoPE)
    glm = kv == 4
    cache, cs, g = _headline_cache(kv, glm, L=65536)
    L = _lib.lib()
    L.skv_debug_rb_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    S = cache.select_sets
    q = (torch.randn(1, 32, 1, 128, device="cuda:0", generator=g) * 1.5).bfloat16()
    kv_len = cache.sparse_end + 3
    lm_idx = cache.k_landmark_idx[0][0].cpu()
    gc = torch.Generator().manual_seed(3)
    buf = np.zeros(16 * 128 * 8, dtype=np.uint64)
    for it in range(4):
        cache.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        torch.cuda.synchronize()
        pos = cache.position_ids[0][0].cpu().clone()
        n_replace = S - int(round(hit * S))
        for h in range(kv):
            pool = torch.tensor(sorted(set(lm_idx[h].tolist()) - set(pos[h].tolist())))
            slots = torch.randperm(S, generator=gc)[:n_replace]
            pos[h, slots] = pool[torch.randperm(len(pool), generator=gc)[:n_replace]]
        cache.position_ids[0][0].copy_(pos.to("cuda:0"))
        torch.cuda.synchronize()
        L.skv_debug_rb_stamps(buf.ctypes.data, 1)
        cache.select_fetch_attend_inplace(0, q, cs, kv_len=kv_len)
        torch.cuda.synchronize()
        L.skv_debug_rb_stamps(buf.ctypes.data, 1)
        st = buf.reshape(16, 128, 8).astype(np.int64)
        t0 = st[st > 0].min()
        tiles = st[:kv, :32]; att = st[:kv, 32:32 + cache._overlap_splits()]
        live = tiles[..., 0] > 0
        rel = lambda a: (a - t0) / 100.0
        names = ["start", "host loads issued", "K tile ready", "weights done", "V arrived", "V image + barrier", "end"]
        print(f"run {it}: {kv} KV heads, hit {hit}: {int(live.sum())} live tiles; kernel span {rel(st.max()):.2f} us")
        for i, n in enumerate(names):
            v = rel(tiles[..., i][live])
            print(f"   tile WGs  {n:<18} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
        a0, a1 = rel(att[..., 0]), rel(att[..., 7])
        print(f"   resident-row attention WGs: start {a0.min():.2f}..{a0.max():.2f}, end {a1.min():.2f}..{a1.max():.2f}")

main()
