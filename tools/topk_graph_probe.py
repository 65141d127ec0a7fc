"""Probe (pure PyTorch, none of this repo's kernels): does torch.topk over [bs, 128256] logits survive
hipGraph capture + replay for bs > 1?  Used to localise the bs > 1 graph fault."""
import sys
import torch
import torch.nn.functional as F

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
what = sys.argv[2] if len(sys.argv) > 2 else "topk"
dev = "cuda:0"
logits = torch.randn(bs, 128256, device=dev)
out = torch.zeros(bs, 1, dtype=torch.long, device=dev)


def body():
    if what == "topk":
        vals, idx = torch.topk(logits / 0.6, 50, dim=-1)
        probs = F.softmax(vals, dim=-1)
        pick = torch.argmax(probs / torch.empty_like(probs).exponential_(1.0), dim=-1, keepdim=True)
        out.copy_(idx.gather(-1, pick))
    else:
        out.copy_((logits / 0.6).argmax(dim=-1, keepdim=True))


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body(); body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    body()
print("captured", flush=True)
for i in range(4):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, out.flatten().tolist(), flush=True)
print("OK", bs, what)
