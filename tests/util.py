"""Shared helpers for the parity tests (test infrastructure)."""
import math

import numpy as np
import torch

ALPHA = 1.0 / math.sqrt(128)


def bits(t):
    """bf16 tensor -> int32 tensor of the raw 16-bit patterns (for exact comparisons)."""
    return t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF


def assert_bits_equal(a, b, what=""):
    a, b = bits(a.cpu()), bits(b.cpu())
    bad = (a != b).nonzero()
    assert bad.numel() == 0, f"{what}: {bad.shape[0]} of {a.numel()} bf16 values differ, first at {bad[0].tolist()}"


def ulp_diff_bf16(a, b):
    """distance in bf16 ulps between two bf16 tensors (sign-magnitude -> monotone integer)."""
    def key(t):
        v = bits(t.cpu()).to(torch.int64)
        return torch.where(v >= 0x8000, 0x8000 - v, v)
    return (key(a) - key(b)).abs()


def make_selection_step(gen, n_chunks, S, hit_frac, cached=None):
    """Random resident set + new selection with a given hit fraction (distinct ids)."""
    if cached is None:
        cached = torch.randperm(n_chunks, generator=gen)[:S]
    n_hit = int(round(hit_frac * S))
    keep = cached[torch.randperm(S, generator=gen)[:n_hit]]
    pool = torch.ones(n_chunks, dtype=torch.bool)
    pool[cached] = False
    rest = pool.nonzero().flatten()
    new = rest[torch.randperm(rest.numel(), generator=gen)[: S - n_hit]]
    cur = torch.cat([keep, new])[torch.randperm(S, generator=gen)]
    return cached.to(torch.int64), cur.to(torch.int64)
