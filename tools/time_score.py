"""Event-timed landmark scan (skv_score_tile_kernel) at the headline shape, cycling 32 tables (1 GB)."""
import math, sys, torch
sys.path.insert(0, ".")
from shadowkv_amd import _lib
L = _lib.lib(); dev = "cuda:0"
for (B, G, N) in ((8, 4, 15560), (4, 8, 25544)):
    T = (N + 255) // 256
    tabs = [torch.randn(B, N, 128, device=dev).bfloat16() for _ in range(32)]
    q = torch.randn(B, G, 128, device=dev).bfloat16()
    D = torch.empty(B, G, N, device=dev, dtype=torch.bfloat16); pm = torch.empty(B, T, G, device=dev); ps = torch.empty(B, T, G, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        for t in tabs:
            L.skv_score_landmarks(q.data_ptr(), t.data_ptr(), D.data_ptr(), pm.data_ptr(), ps.data_ptr(), B, G, N, 1 / math.sqrt(128), st)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [run() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * 32)
    print(f"B={B} G={G} N={N}: {us:.2f} us/launch  {B*N*256/us*1e-6:.2f} TB/s")
