"""How much would overlapping a layer's QKV GEMV (50 MB of weights) with the landmark scan (31.9 MB) buy?  The scan needs q, the
GEMV's result - but only for its dot products, not for its loads.  Upper bound of a fused launch (GEMV role + scan role with a
hand-off of q): the two launches issued on TWO streams at the same time, without the dependency, against the same two launches
back to back on one stream (what the step does).  Cycles 32 weight matrices / landmark tables (far beyond the caches)."""
import math, sys, torch
sys.path.insert(0, ".")
from shadowkv_amd import _lib
L = _lib.lib(); dev = "cuda:0"
B, G, N = 8, 4, 15560
tabs = [torch.randn(B, N, 128, device=dev).bfloat16() for _ in range(32)]
ws_ = [(torch.randn(6144, 4096, device=dev) * 0.02).bfloat16() for _ in range(32)]
q = torch.randn(B, G, 128, device=dev).bfloat16()
x = torch.randn(4096, device=dev).bfloat16()
y = torch.empty(6144, device=dev, dtype=torch.bfloat16)
ws = torch.empty(L.skv_select_workspace_bytes(B, G, N), dtype=torch.uint8, device=dev)
state = torch.zeros(L.skv_select_state_bytes(B, G), dtype=torch.uint8, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def gemv(i, st):
    _lib.check(L.skv_gemv_bf16(ws_[i].data_ptr(), x.data_ptr(), 0, y.data_ptr(), 6144, 4096, 0, st), "gemv")


def scan(i, st):
    _lib.check(L.skv_score_landmarks_fused(q.data_ptr(), tabs[i].data_ptr(), 0, ws.data_ptr(), B, G, N, 1 / math.sqrt(128),
                                           state.data_ptr(), 0, 0, 0, st), "scan")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 32)


cur = torch.cuda.current_stream().cuda_stream
print(f"GEMV 6144 x 4096 alone          {timed(lambda: [gemv(i, cur) for i in range(32)]):6.2f} us")
print(f"scan alone                      {timed(lambda: [scan(i, cur) for i in range(32)]):6.2f} us")
print(f"GEMV then scan, one stream      {timed(lambda: [(gemv(i, cur), scan(i, cur)) for i in range(32)]):6.2f} us per pair")


def both():
    # pair i on two streams at the same time; the pairs are serialised by events so that pair i + 1 starts when both are done
    main = torch.cuda.current_stream()
    for i in range(32):
        ev = torch.cuda.Event(); ev.record(main)
        sa.wait_event(ev); sb.wait_event(ev)
        gemv(i, sa.cuda_stream); scan(i, sb.cuda_stream)
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(sa); eb.record(sb)
        main.wait_event(ea); main.wait_event(eb)


print(f"GEMV || scan, two streams       {timed(both):6.2f} us per pair (includes two eager cross-stream edges per pair: not an upper bound of anything)")


def free_running(reps=5):
    """160 GEMVs on one stream and 160 scans on another, started together, no dependency between them: what the chip does with
    both kinds of work in flight (an upper bound for a fused launch whose scan role never waits for q)."""
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    sa.wait_event(e0); sb.wait_event(e0)
    for _ in range(reps):
        for i in range(32):
            gemv(i, sa.cuda_stream); scan(i, sb.cuda_stream)
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa); eb.record(sb)
    main.wait_event(ea); main.wait_event(eb)
    e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 32)


free_running(1)
print(f"GEMVs and scans free-running on two streams: {free_running():6.2f} us per pair")
