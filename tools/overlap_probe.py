"""Can a background PCIe fetch (score-guided prefetch of chunks the NEXT step is likely to select) ride
under the dense GEMVs for free?  Stream A: gate/up-shaped GEMVs over HBM-cold weights.  Stream B: a host -> HBM gather of
`chunks` 2-KiB rows per head (skv_gather_copy: plain in-kernel loads over PCIe).  Reports each alone and both together."""
import sys, torch
sys.path.insert(0, ".")
from shadowkv_amd import _lib
from shadowkv_amd.kv_cache import pinned_host_tensor
L = _lib.lib(); dev = "cuda:0"
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 128
heads, table = 8, 15616
N, K = 28672, 4096
ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(8)]
x = torch.randn(1, K, device=dev).bfloat16(); y = torch.empty(1, N // 2, device=dev, dtype=torch.bfloat16)
vhost = pinned_host_tensor((1, heads, table, 1024), torch.bfloat16); vhost.normal_()
vdev = torch.zeros(1, heads, chunks, 1024, device=dev, dtype=torch.bfloat16)
ids = torch.stack([torch.randperm(table)[:chunks] for _ in range(heads)]).view(1, heads, chunks).to(dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
REP = 64
def gemvs():
    for i in range(REP):
        L.skv_gemv_bf16(ws[i % 8].data_ptr(), x.data_ptr(), 0, y.data_ptr(), N, K, 1, sa.cuda_stream)
def fetches():
    for i in range(REP):
        L.skv_gather_copy(vhost.data_ptr(), vdev.data_ptr(), ids.data_ptr(), 1, heads, table * 1024, chunks * 1024, chunks, sb.cuda_stream)
def timed(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sa.wait_event(e0); sb.wait_event(e0)
    if fa: fa()
    if fb: fb()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa); eb.record(sb)
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP
for _ in range(2): timed(gemvs, fetches)
g = timed(gemvs, None); f = timed(None, fetches); both = timed(gemvs, fetches)
mb = heads * chunks * 2048 / 1e6
print(f"{chunks} chunks/head ({mb:.2f} MB per fetch): GEMV alone {g:.1f} us | fetch alone {f:.1f} us ({mb / f * 1e3:.1f} GB/s) | "
      f"both streams {both:.1f} us per pair (serial would be {g + f:.1f}, perfect overlap {max(g, f):.1f})", flush=True)
