#!/usr/bin/env python3
"""In-kernel phase stamps of the sampler launch (skv_sample_topk_kernel: exact top-k of 128,256 bf16 logits, top-p, draw).
Build the diagnostic library first: make -C shadowkv_amd/csrc stamps ; run with SKV_LIB_PATH=shadowkv_amd/libshadowkv_hip_stamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from shadowkv_amd import _lib, tensor_op

def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 128256
    ranges = len(sys.argv) > 2 and sys.argv[2] == "ranges"
    L = _lib.lib()
    L.skv_debug_sample_stamps.argtypes = [ctypes.c_void_p]
    g = torch.Generator(device="cuda:0").manual_seed(1)
    buf = np.zeros(32, dtype=np.uint64)
    flush = torch.empty(640 << 20, dtype=torch.uint8, device="cuda:0")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(6):
        logits = (torch.randn(1, V, device="cuda:0", generator=g) * 3.0).bfloat16()
        flush.fill_(it)
        torch.cuda.synchronize()
        e0.record()
        rm = None
        if ranges:     # the keys the lm_head launch would have left (computed with torch here)
            b = logits.view(torch.int16).to(torch.int32).view(-1) & 0xffff
            keys = torch.where((b & 0x8000) != 0, (~b) & 0xffff, b | 0x8000).view(V // 16, 16).max(dim=-1).values
            rm = torch.zeros(1, (V // 16 + 7) // 8 * 8, dtype=torch.int16, device="cuda:0")
            rm[0, :V // 16] = keys.to(torch.int16)
            flush.fill_(it + 100)
            torch.cuda.synchronize()
            e0.record()
        tok = tensor_op.sample_token_native(logits, 0.6, 50, 0.9, seed=7, range_max=rm)
        e1.record()
        torch.cuda.synchronize()
        L.skv_debug_sample_stamps(buf.ctypes.data)
        st = buf.astype(np.int64)
        r = lambda i: (st[i] - st[19]) / 100.0
        print(f"run {it}: token {int(tok)} | events {e0.elapsed_time(e1) * 1e3:.1f} us | row in registers {r(20):.2f} bound (k-th of the thread maxima) {r(21):.2f} "
              f"candidates compacted {r(22):.2f} exact k-th {r(23):.2f} part done {r(27):.2f} merged {r(28):.2f} drawn {r(29):.2f}")

main()
