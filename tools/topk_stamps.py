#!/usr/bin/env python3
"""In-kernel phase stamps of the fused top-k launch INSIDE the decode step's launch sequence (early fetch on: selection
workgroups + pull workgroups), at the headline shape: where the launch's 17 us go.
Build the diagnostic library first: make -C shadowkv_amd/csrc stamps ; run with SKV_LIB_PATH=shadowkv_amd/libshadowkv_hip_stamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
from shadowkv_amd import _lib
from test_gpu_kv_cache import _headline_cache

def main():
    kv = int(sys.argv[1]) if len(sys.argv) > 1 else 8            # 8: Llama (G = 4); 4: GLM-4 (G = 8, GLM RoPE)
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cache, cs, g = _headline_cache(kv, kv == 4, L=124928 if kv == 8 else 204800)
    cache.enable_early_fetch(early_max=E or None)
    L = _lib.lib()
    L.skv_debug_topk_stamps.argtypes = [ctypes.c_void_p]
    q32 = torch.randn(1, 32, 1, 128, device="cuda:0", generator=g) * 1.5
    kv_len = cache.sparse_end + 3
    buf = np.zeros(32, dtype=np.uint64)
    flush = torch.empty(640 << 20, dtype=torch.uint8, device="cuda:0")
    for it in range(8):
        q32 = q32 + 0.35 * torch.randn(q32.shape, device="cuda:0", generator=g)
        flush.fill_(it)                                           # the step's other layers: caches cold as in the real step
        torch.cuda.synchronize()
        cache.select_fetch_attend_inplace(0, q32.bfloat16(), cs, kv_len=kv_len)
        torch.cuda.synchronize()
        L.skv_debug_topk_stamps(buf.ctypes.data)
        st = buf.astype(np.int64)
        t0 = min(st[0], st[24])
        r = lambda i: (st[i] - t0) / 100.0
        misses = int(cache.block_num * cache.select_sets - int(cache.cnts.sum()))
        print(f"step {it}: misses {misses} pulled {int(cache.early_fetch_counts(0).sum())} | select WG: start {r(0):.2f} finals {r(12):.2f} level {r(13):.2f} "
              f"(search {r(14) - r(13):.2f}) gather+exact {r(16):.2f} search2 {r(17):.2f} placed {r(18):.2f} ids {r(6):.2f} lookup {r(7):.2f} "
              f"scan3 {r(8):.2f} vote {r(9):.2f} end {r(10):.2f} | pull WG 0: start {r(24):.2f} [resident ids in {r(19):.2f} slot ids {r(20):.2f} "
              f"barrier {r(21):.2f} clears acked {r(22):.2f} scan {r(23):.2f}] list ready {r(25):.2f} pulled {r(26):.2f}")

main()
