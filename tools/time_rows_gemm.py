"""Event-timed small-M projection kernel (skv_linear_rows_bf16) at the decode shapes, cycling 8 weight copies."""
import sys, torch
sys.path.insert(0, ".")
from shadowkv_amd import _lib
L = _lib.lib(); dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for (N, K, silu) in ((6144, 4096, 0), (4096, 4096, 0), (28672, 4096, 1), (4096, 14336, 0)):
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(8)]
    x = torch.randn(M, K, device=dev).bfloat16()
    y = torch.empty(M, N // 2 if silu else N, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        for w in ws:
            L.skv_linear_rows_bf16(w.data_ptr(), x.data_ptr(), 0, y.data_ptr(), M, N, K, silu, st)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [run() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 80
    print(f"M={M} N={N} K={K} silu={silu}: {us:.2f} us  {N*K*2/us*1e-6:.2f} TB/s", flush=True)
