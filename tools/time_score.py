"""Event-timed landmark scan (skv_score_tile_kernel) at the headline shape, cycling 32 tables (1 GB).
SCAN_FORM=fused (default): the launch of the fused selection (keys + slot-major logits, as the decode step issues it since
round 4); SCAN_FORM=plain: the three-launch form (logits [B][G][N])."""
import math, os, sys, torch
sys.path.insert(0, ".")
from shadowkv_amd import _lib
L = _lib.lib(); dev = "cuda:0"
form = os.environ.get("SCAN_FORM", "fused")
for (B, G, N) in ((8, 4, 15560), (4, 8, 25544)):
    T = (N + 255) // 256
    tabs = [torch.randn(B, N, 128, device=dev).bfloat16() for _ in range(32)]
    q = torch.randn(B, G, 128, device=dev).bfloat16()
    D = torch.empty(B, G, N, device=dev, dtype=torch.bfloat16); pm = torch.empty(B, T, G, device=dev); ps = torch.empty(B, T, G, device=dev)
    ws = torch.empty(L.skv_select_workspace_bytes(B, G, N), dtype=torch.uint8, device=dev)
    state = torch.zeros(L.skv_select_state_bytes(B, G), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        for t in tabs:
            if form == "fused":
                _lib.check(L.skv_score_landmarks_fused(q.data_ptr(), t.data_ptr(), 0, ws.data_ptr(), B, G, N, 1 / math.sqrt(128),
                                                       state.data_ptr(), 0, 0, 0, st), "scan")
            else:
                L.skv_score_landmarks(q.data_ptr(), t.data_ptr(), D.data_ptr(), pm.data_ptr(), ps.data_ptr(), B, G, N, 1 / math.sqrt(128), st)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [run() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * 32)
    print(f"{form} B={B} G={G} N={N}: {us:.2f} us/launch  {B*N*256/us*1e-6:.2f} TB/s")
