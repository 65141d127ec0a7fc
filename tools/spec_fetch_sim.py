#!/usr/bin/env python3
"""How well does 'score >= the previous step's k-th value, and not resident' predict the chunks a decode step will miss?
(issue the host loads of the certain misses before the top-k has finished.)  Simulation on the bench's
own workload: Llama-3.1-8B landmarks at 122K (synthetic context, a few layers), the query walk of bench.py; scores computed
with torch in f32 (softmax over the landmarks per query head, maximum over the GQA group, bf16) - close enough to the
kernel's values to count sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import math
import torch
from shadowkv_amd import llama


def main():
    layers = 3
    m = llama.DecoderLM(cfg=llama.LLAMA_3_1_8B, batch_size=1, max_length=122 * 1024, device="cuda:0", sparse_budget=2048,
                        rank=160, chunk_size=8, num_layers=layers, chunk_layout="inplace", overlap_attention=True)
    llama.build_synthetic_context(m, 122 * 1024, seed=4321)
    c = m.kv_cache
    steps = 40
    table = llama.make_walk_table(m, steps, step=0.3, seed=1234)       # [steps, L, 1, Hq, 1, D]
    S, G, Hkv = c.select_sets, m.num_heads // m.num_key_value_heads, m.num_key_value_heads
    # (b) the same prediction made BEFORE the normalisation exists, in logit space with the previous step's statistics:
    # flag slot n if for some query head g  D[g][n] >= m_prev[g] + ln(thr_prev / inv_prev[g]) + margin
    for margin in (0.0, 0.02, 0.05, -0.02):
        tot = dict(miss=0, flagged=0, good=0, sel=0, cap=0)
        for l in range(layers):
            lm = c.k_landmark[l][0].float()
            prev = None
            for t in range(steps):
                q = table[t, l, 0, :, 0].float().view(Hkv, G, -1)
                logits = (torch.einsum("hgd,hnd->hgn", q, lm) / math.sqrt(128)).bfloat16().float()
                mx = logits.max(dim=-1, keepdim=True).values
                ssum = torch.exp(logits - mx).sum(dim=-1, keepdim=True)
                p = (torch.exp(logits - mx) / ssum).bfloat16()
                score = p.max(dim=1).values.float()
                top = torch.topk(score, S, dim=-1)
                sel = torch.zeros_like(score, dtype=torch.bool).scatter_(1, top.indices, True)
                thr = top.values[:, -1]
                if prev is not None and t >= 4:
                    psel, pthr, pmx, psum = prev
                    dthr = pmx + torch.log(pthr[:, None, None] * psum) + margin          # [Hkv, G, 1]
                    flagged = (logits >= dthr).any(dim=1) & ~psel
                    miss = sel & ~psel
                    tot["miss"] += int(miss.sum()); tot["flagged"] += int(flagged.sum())
                    tot["good"] += int((flagged & miss).sum()); tot["sel"] += int(sel.sum())
                    tot["cap"] = max(tot["cap"], int(((logits >= dthr).any(dim=1)).view(Hkv, -1, 1).sum(dim=1).max()))
                prev = (sel, thr, mx, ssum)
        print(f"logit-space flag, margin {margin:+.2f}: flagged early {tot['good'] / tot['miss']:.3f} of the misses; wasted "
              f"{(tot['flagged'] - tot['good']) / tot['miss']:.3f} of the miss bytes; most flagged slots (resident ones included) in one head and step: {tot['cap']}")
    for margin in (1.0, 1.02, 1.05, 0.98):
        tot = dict(miss=0, flagged=0, good=0, sel=0)
        for l in range(layers):
            lm = c.k_landmark[l][0].float()                             # [Hkv, N, D]
            prev_sel, prev_thr = None, None
            for t in range(steps):
                q = table[t, l, 0, :, 0].float().view(Hkv, G, -1)       # [Hkv, G, D]
                logits = torch.einsum("hgd,hnd->hgn", q, lm) / math.sqrt(128)
                p = torch.softmax(logits.bfloat16().float(), dim=-1).bfloat16()
                score = p.max(dim=1).values.float()                     # [Hkv, N]
                top = torch.topk(score, S, dim=-1)
                sel = torch.zeros_like(score, dtype=torch.bool).scatter_(1, top.indices, True)
                thr = top.values[:, -1]
                if prev_sel is not None and t >= 4:
                    miss = sel & ~prev_sel
                    flagged = (score >= prev_thr[:, None] * margin) & ~prev_sel
                    tot["miss"] += int(miss.sum()); tot["flagged"] += int(flagged.sum())
                    tot["good"] += int((flagged & miss).sum()); tot["sel"] += int(sel.sum())
                prev_sel, prev_thr = sel, thr
        print(f"flag threshold = {margin:.2f} x previous k-th value: misses {tot['miss'] / tot['sel']:.3f} of the selection; "
              f"flagged early {tot['good'] / tot['miss']:.3f} of the misses; wasted fetches {1 - tot['good'] / max(tot['flagged'], 1):.3f} "
              f"of the flagged ({(tot['flagged'] - tot['good']) / tot['miss']:.3f} of the miss bytes)")


main()
