#!/usr/bin/env python3
"""Does a decode GEMV run faster when (part of) its weights were read into the 256 MiB Infinity Cache just before?
The fetch launch is bound by the PCIe link and the selection chain by latency: HBM idles for ~55 us per layer in front
of o_proj / gate-up.  This probe times the native GEMV cold (1 GiB of other traffic in front), after a plain streaming
read of the first X MB of its weights, and replayed back to back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from shadowkv_amd import tensor_op


def timed(fn, before, reps=12):
    ts = []
    for _ in range(reps):
        before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(1)
    flush = torch.empty(1 << 29, dtype=torch.int16, device=dev).fill_(1)        # 1 GiB
    for name, N, K, fuse in (("o_proj 4096x4096", 4096, 4096, False), ("gate/up 28672x4096 (silu-mul)", 28672, 4096, True),
                             ("down 4096x14336", 4096, 14336, False)):
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        x = torch.randn(1, K, device=dev, generator=g).bfloat16()
        out = torch.empty(1, N // 2 if fuse else N, dtype=torch.bfloat16, device=dev)
        mb = w.numel() * 2 / 1e6
        run = lambda: tensor_op.linear_decode(x, w, None, fuse, out=out)
        wi = w.view(torch.int16).view(-1)
        run(); torch.cuda.synchronize()
        cold = timed(run, lambda: flush.sum())
        print(f"{name}: {mb:.1f} MB")
        print(f"   cold (1 GiB of other reads in front)      median {cold[0]:7.2f} us  min {cold[1]:7.2f}  ({mb / cold[0]:.2f} TB/s)")
        for frac in (0.25, 0.5, 0.75, 1.0):
            n = int(wi.numel() * frac)
            if n * 2 > 230e6:
                n = int(230e6 / 2)
            def pre():
                flush.sum(); wi[:n].sum()
            t = timed(run, pre)
            print(f"   first {n * 2 / 1e6:6.1f} MB read in front (plain loads)  median {t[0]:7.2f} us  min {t[1]:7.2f}  ({mb / t[0]:.2f} TB/s)")
        warm = timed(run, lambda: None)
        print(f"   replayed back to back                     median {warm[0]:7.2f} us  min {warm[1]:7.2f}  ({mb / warm[0]:.2f} TB/s)")
        del w


main()
