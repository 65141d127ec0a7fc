#!/usr/bin/env python3
"""(Record of a rejected experiment: needs the library built WITH tools/rejected/merge_oproj_fused.hip and its tensor_op wrappers,
see profiles/r03_merge_oproj_fused.txt.)  The fused merge + O projection launch against the two launches it replaces: time per call (20 captured back to back over
4 weight sets) and, with the diagnostic library (make -C shadowkv_amd/csrc stamps; SKV_LIB_PATH=.../libshadowkv_hip_stamps.so),
in-kernel phase stamps."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from shadowkv_amd import tensor_op
from shadowkv_amd._lib import lib, check, ptr, current_stream_handle


def main():
    dev, Hq, H, Hkv, splits, S = "cuda:0", 32, 4096, 8, 32, 256
    g = torch.Generator(device=dev).manual_seed(1)
    sets = [(torch.randn(H, H, device=dev, generator=g) * 0.03).bfloat16() for _ in range(4)]
    flush = [(torch.randn(4096, 14336, device=dev, generator=g) * 0.03).bfloat16() for _ in range(2)]   # 117 MB each, between calls
    xf = torch.randn(1, 1, 14336, device=dev, generator=g).bfloat16()
    nrec = splits + S // 8
    ws = torch.randn(Hq, nrec, 132, device=dev, generator=g)
    ws[..., 129] = ws[..., 129].abs() + 1
    cnts = torch.full((Hkv,), int(0.67 * S), device=dev, dtype=torch.int32)
    sync = tensor_op.merge_oproj_workspace(dev)

    def fused(i):
        tensor_op.linear_decode(xf, flush[i % 2])            # something HBM-bound in front, as in the decode step
        return tensor_op.attention_merge_oproj(ws, cnts, Hq, Hkv, S, splits, sets[i % 4], sync)

    def two(i):
        tensor_op.linear_decode(xf, flush[i % 2])
        attn = torch.empty(1, 1, Hq, 128, dtype=torch.bfloat16, device=dev)
        check(lib().skv_attn_finish_inplace(ptr(ws), ptr(cnts), ptr(attn), 1, Hq, Hkv, S, splits, current_stream_handle()), "f")
        return tensor_op.linear_decode(attn.reshape(1, 1, H), sets[i % 4])

    def only_flush(i):
        tensor_op.linear_decode(xf, flush[i % 2])

    res = {}
    for label, fn in (("flush GEMV alone", only_flush), ("two launches", two), ("fused", fused)):
        for i in range(4):
            fn(i)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(20):
                fn(i)
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 20)
        res[label] = min(ts)
        print(f"{label:18s} {min(ts):7.2f} us per call (20 captured back to back, best of 5)")
    print(f"merge + O projection: two launches {res['two launches'] - res['flush GEMV alone']:.2f} us, fused {res['fused'] - res['flush GEMV alone']:.2f} us")
    print("status", tensor_op.merge_oproj_status(sync))
    L = lib()
    if hasattr(L, "skv_debug_mo_stamps"):
        fused(0); torch.cuda.synchronize()
        buf = np.zeros(1024 * 8, dtype=np.uint64)
        L.skv_debug_mo_stamps.argtypes = [ctypes.c_void_p]
        L.skv_debug_mo_stamps(buf.ctypes.data)
        st = buf.reshape(1024, 8)[:Hq + H // 8].astype(np.int64)
        t0 = st[:, 0].min()
        rel = (st - t0) / 100.0
        m, gv = rel[:Hq], rel[Hq:]
        for n, col in (("start", 0), ("merged (stores acknowledged)", 1), ("arrived (release + add acknowledged)", 2)):
            print(f"   merge WGs {n:38s} min {m[:, col].min():6.2f} median {np.median(m[:, col]):6.2f} max {m[:, col].max():6.2f}")
        for n, col in (("start", 0), ("weights requested", 3), ("poll done", 4), ("x here", 5), ("end", 6)):
            print(f"   GEMV  WGs {n:38s} min {gv[:, col].min():6.2f} median {np.median(gv[:, col]):6.2f} max {gv[:, col].max():6.2f}")


main()
