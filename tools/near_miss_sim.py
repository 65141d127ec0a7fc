#!/usr/bin/env python3
"""VERDICT r4 item 4a, before building it: how many of step t+1's misses are among step t's NEAR misses (ranks S+1 .. S+P of
the group-max score)?  Those are the chunks a pull under the gate/up GEMV of step t could have staged for step t+1.
Same workload as tools/spec_fetch_sim.py (Llama-3.1-8B landmarks at 122K, synthetic context, bench.py's query walk, torch f32
scores rounded like the kernel's)."""
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from shadowkv_amd import llama


def main():
    layers, steps = 3, 40
    walk_step = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    m = llama.DecoderLM(cfg=llama.LLAMA_3_1_8B, batch_size=1, max_length=122 * 1024, device="cuda:0", sparse_budget=2048,
                        rank=160, chunk_size=8, num_layers=layers, chunk_layout="inplace", overlap_attention=True)
    llama.build_synthetic_context(m, 122 * 1024, seed=4321)
    c = m.kv_cache
    table = llama.make_walk_table(m, steps, step=walk_step, seed=1234)
    S, G, Hkv = c.select_sets, m.num_heads // m.num_key_value_heads, m.num_key_value_heads
    buckets = [(0, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 512)]
    enter = {b: [0, 0] for b in buckets}
    cover = {P: [0, 0, 0] for P in (16, 32, 64, 128, 256)}          # useful, pulled, misses
    # the in-step flag (score >= previous k-th value, not resident) pulls up to E per head in the top-k launch: what is left for
    # the fetch launch with and without the near-miss staging
    sel_tot = miss_tot = 0
    for l in range(layers):
        lm = c.k_landmark[l][0].float()
        prev = None
        for t in range(steps):
            q = table[t, l, 0, :, 0].float().view(Hkv, G, -1)
            logits = (torch.einsum("hgd,hnd->hgn", q, lm) / math.sqrt(128)).bfloat16().float()
            p = torch.softmax(logits, dim=-1).bfloat16()
            score = p.max(dim=1).values.float()
            order = torch.argsort(score, dim=-1, descending=True, stable=True)
            sel = torch.zeros_like(score, dtype=torch.bool).scatter_(1, order[:, :S], True)
            if prev is not None and t >= 4:
                psel, porder = prev
                miss = sel & ~psel
                sel_tot += int(sel.sum()); miss_tot += int(miss.sum())
                for (a, b) in buckets:
                    idx = porder[:, S + a:S + b]
                    enter[(a, b)][0] += int(sel.gather(1, idx).sum()); enter[(a, b)][1] += idx.numel()
                for P in cover:
                    idx = porder[:, S:S + P]
                    cover[P][0] += int(sel.gather(1, idx).sum()); cover[P][1] += idx.numel(); cover[P][2] += int(miss.sum())
            prev = (sel, order)
    print(f"walk step {walk_step}: misses {miss_tot / sel_tot:.3f} of the selection ({miss_tot / (layers * (steps - 4) * Hkv):.1f} chunks per head and step)")
    for (a, b), (n, d) in enter.items():
        print(f"  ranks S+{a + 1:3d} .. S+{b:3d} of step t: {n / d:.3f} are selected at step t+1")
    for P, (u, n, mi) in cover.items():
        print(f"  staging the {P:3d} nearest misses per head: {u / n:.3f} of them used next step = {u / mi:.3f} of its misses "
              f"({u / (layers * (steps - 4) * Hkv):.1f} chunks per head)")


main()
