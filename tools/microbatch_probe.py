"""Two half-batches on two streams: does one half's dense / selection work hide under the other half's PCIe-bound fetch?
(At bs >= 8 per GPU the fetch launch is ~65 % of a layer and nothing else runs beside it.)
usage: python tools/microbatch_probe.py <half batch> [resident sets]   -> tokens/s of one half alone and of both interleaved"""
import sys, time, types, torch
sys.path.insert(0, ".")
import bench
from shadowkv_amd import llama, tensor_op
half = int(sys.argv[1]) if len(sys.argv) > 1 else 12
R = int(sys.argv[2]) if len(sys.argv) > 2 else None
steps, warm = 16, 6
args = types.SimpleNamespace(attn="shadowkv", batch=half, layers=None, layout="inplace", v_table="host", overlap_attention=1,
                             warmup=warm, steps=steps, resident_sets=R, mode="graph", query_mode="walk")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
decs, keep = [], []
for r in range(2):
    model, cfg, ctx, budget, tb = bench.build_model("llama31_122k", args, r, dev)
    table = llama.make_walk_table(model, warm + 3 * steps + 8, step=0.3, seed=5 + r)
    dec = llama.GraphDecoder(model, temperature=0.6, walk_table=table, ring_slack=True)
    dec.token.copy_(torch.randint(0, cfg.vocab_size, (half, 1), device=dev))
    dec.capture()
    keep.append(list(tensor_op._attn_ws.values())); tensor_op._attn_ws.clear()      # every graph its own attention workspace
    decs.append(dec)
    print(f"half {r}: bs {half} built in {tb:.0f} s", flush=True)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(which, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for i in which:
            with torch.cuda.stream(streams[i]):
                decs[i].step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run([0, 1], warm)
t_one = run([0], steps); t_both = run([0, 1], steps)
print(f"half batch {half}, resident sets {R or 256}: one half alone {half * steps / t_one:.1f} tok/s ({t_one / steps * 1e3:.2f} ms/step) | "
      f"two halves on two streams {2 * half * steps / t_both:.1f} tok/s ({t_both / steps * 1e3:.2f} ms per step pair)", flush=True)
