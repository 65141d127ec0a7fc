import sys, os, time, math
sys.path.insert(0, "/root/repo")
import torch, oracle
kv, G, N, D, S, C = 8, 4, 15560, 128, 256, 8
g = torch.Generator().manual_seed(0)
q = torch.randn(kv, G, D, generator=g).bfloat16(); lm = torch.randn(kv, N, D, generator=g).bfloat16()
T = (N + 255) // 256
Dm = torch.zeros(kv, G, N, dtype=torch.bfloat16); P = torch.zeros_like(Dm); nm = torch.zeros(kv, T, G); sm = torch.zeros(kv, T, G)
lm_idx = torch.arange(N).repeat(kv, 1).contiguous()
print("threads", oracle.num_threads(), "affinity", len(os.sched_getaffinity(0)))
def t(name, f, n=3):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    print(f"{name:<28} {(time.perf_counter() - t0) / n * 1e3:8.1f} ms")
t("batch_gemm_softmax", lambda: oracle.batch_gemm_softmax(q, lm, Dm, nm, sm, P, kv, G, N, D, 1 / math.sqrt(128)))
t("group_max_topk", lambda: oracle.group_max_topk(P, lm_idx, kv, G, N, S))
L = 124928; U = torch.randn(1, L, 160, generator=g).bfloat16(); SV = torch.randn(1, kv, 128, 160, generator=g).bfloat16()
ids = torch.randint(0, L // 8, (1, kv, S), dtype=torch.int32); pre = torch.zeros(1, kv, S * C, D, dtype=torch.bfloat16)
cnt = torch.full((kv,), 172, dtype=torch.int32)
t("batch_gather_gemm", lambda: oracle.batch_gather_gemm(U, SV, None, None, ids, pre, 1, kv, L, D, 160, S * C, 0, C, cnt))
k = torch.randn(1, kv, 2592, D, generator=g).bfloat16(); v = torch.randn(1, kv, 2592, D, generator=g).bfloat16()
t("sparse_attention", lambda: oracle.sparse_attention(q.view(1, 32, D), k, v, 2497, 1 / math.sqrt(D)))
