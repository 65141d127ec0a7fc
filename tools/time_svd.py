"""Times the two rank-160 factorisations of ShadowKVCache_CPU.get_svd on one layer's pre-RoPE keys at the headline
context (K [1, 124928, 1024] f32 built from bf16 values: rank-256 signal + noise, so the spectrum has a tail):
torch.svd (the reference's call, rocSOLVER) against svd_mode='gram' (K^T K eigh + two GEMMs).  Prints wall time and
the relative error of the rank-160 reconstruction from the bf16-stored factors."""
import sys
import time
import torch
from shadowkv_amd.kv_cache import gram_factorize

L = int(sys.argv[1]) if len(sys.argv) > 1 else 124928
which = sys.argv[2] if len(sys.argv) > 2 else "both"
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(3)
a = torch.randn(1, L, 256, device=dev, generator=g)
b = torch.randn(1, 256, 1024, device=dev, generator=g) * torch.logspace(0, -2, 256, device=dev).view(1, 256, 1)
k = (a @ b + 0.01 * torch.randn(1, L, 1024, device=dev, generator=g)).bfloat16().float()
scale = k.pow(2).mean().sqrt()
r = 160


def err(u, sv):
    return float(((u.bfloat16().float() @ sv.bfloat16().float()) - k).pow(2).mean().sqrt() / scale)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


if which in ("both", "gram"):
    t, (u, sv) = timed(lambda: gram_factorize(k, r), 3)
    print(f"gram  L={L}: {t*1e3:9.1f} ms   rel. reconstruction error {err(u, sv):.5f}", flush=True)
if which in ("both", "svd"):
    def ref():
        u, s, v = torch.svd(k)
        return u[:, :, :r], torch.diag_embed(s[:, :r]) @ v.transpose(1, 2)[:, :r]
    t, (u, sv) = timed(ref, 1)
    print(f"svd   L={L}: {t*1e3:9.1f} ms   rel. reconstruction error {err(u, sv):.5f}", flush=True)
