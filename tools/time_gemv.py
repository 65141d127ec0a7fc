#!/usr/bin/env python3
"""Back-to-back timing of the decode step's dense kernels over 32 layers of distinct weights (HBM-cold, like the step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from shadowkv_amd import tensor_op, llama
dev = "cuda:0"
L = 32
g = torch.Generator(device=dev).manual_seed(0)
def w(n, k): return [(torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16() for _ in range(L)]
wqkv, wo, wgu, wd = w(6144, 4096), w(4096, 4096), w(28672, 4096), w(4096, 14336)
x = torch.randn(1, 1, 4096, device=dev, generator=g).bfloat16(); res = torch.randn(1, 1, 4096, device=dev, generator=g).bfloat16()
xi = torch.randn(1, 1, 14336, device=dev, generator=g).bfloat16()
nw = torch.ones(4096, device=dev, dtype=torch.bfloat16)
cs = llama.build_cos_sin_cache(llama.LLAMA_3_1_8B, 4096, torch.device(dev), torch.bfloat16)
pos = torch.tensor([[100]], device=dev); row = torch.tensor([5], device=dev)
kc = torch.zeros(1, 8, 64, 128, device=dev, dtype=torch.bfloat16); vc = torch.zeros_like(kc)
def t(name, fn, mb):
    s_ = torch.cuda.Stream()
    with torch.cuda.stream(s_):
        for l in range(L): fn(l)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s_):          # captured: no host launch gaps, like the decode step
        for l in range(L): fn(l)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * L)
    print(f"{name:<34} {us:7.2f} us  {mb / us * 1e-0:7.1f} GB/s" .replace("GB/s", "MB/us = %.2f TB/s" % (mb / us)))
t("qkv (norm + gemv + rope/append)", lambda l: tensor_op.norm_qkv_rope_update(x, res, nw, 1e-5, wqkv[l], None, cs, pos, row, kc, vc, 32, 8), 50.33)
t("qkv plain gemv", lambda l: tensor_op.linear_decode(x, wqkv[l]), 50.33)
t("o gemv", lambda l: tensor_op.linear_decode(x, wo[l]), 33.55)
t("gate/up (norm + gemv + silu)", lambda l: tensor_op.norm_linear_decode(x, res, nw, 1e-5, wgu[l], fuse_silu_mul=True), 234.9)
t("gate/up plain gemv + silu", lambda l: tensor_op.linear_decode(x, wgu[l], fuse_silu_mul=True), 234.9)
t("down gemv", lambda l: tensor_op.linear_decode(xi, wd[l]), 117.4)
