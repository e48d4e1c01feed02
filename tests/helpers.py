"""Shared test helpers (CPU side): golden decoding + synthetic paged-cache construction."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def bf16_from_u16(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16).copy()).view(torch.bfloat16)


def fp8_from_u8(a):
    return torch.from_numpy(np.ascontiguousarray(a).copy()).view(torch.float8_e4m3fn)


def rel_mae(x, ref):
    x, ref = x.double(), ref.double()
    return float((x - ref).abs().mean() / ref.abs().mean().clamp_min(1e-30))


def make_paged_case(lens, H, s_q=1, seed=0, extra_pages=3, poison_tail=True, kscale_spread=False):
    """Synthetic per-token-FP8 paged cache on CPU built with the ORACLE's K5 (memory_pool.py:873-880 restated),
    pages randomly permuted (page 0 = padding page, unused), optional NaN-pattern poison past each sequence end."""
    from oracle import mla_ref

    g = torch.Generator().manual_seed(seed)
    bs = len(lens)
    npg = [(max(L, 0) + 63) // 64 for L in lens]
    total_pages = sum(npg) + 1 + extra_pages
    slots = total_pages * 64
    k_lora = torch.zeros(slots, 1, 512, dtype=torch.uint8)
    k_scale = torch.ones(slots, 1, 1, dtype=torch.float32)
    k_rope = torch.zeros(slots, 1, 64, dtype=torch.bfloat16)
    if poison_tail:
        k_lora.fill_(0x7F)  # e4m3fn NaN pattern everywhere; valid tokens are overwritten below
        k_scale.fill_(float("nan"))
        k_rope.fill_(float("nan"))
    perm = (torch.randperm(total_pages - 1, generator=g) + 1).tolist()
    max_pages = max(max(npg), 1) + 2
    block_table = torch.zeros(bs, max_pages, dtype=torch.int32)
    pi = 0
    for b in range(bs):
        for j in range(npg[b]):
            block_table[b, j] = perm[pi]
            pi += 1
        L = max(lens[b], 0)
        if L == 0:
            continue
        t = torch.arange(L)
        loc = (block_table[b, (t // 64).long()].long() * 64 + t % 64).to(torch.int32)
        key = torch.randn(L, 1, 576, generator=g)
        if kscale_spread:
            key = key * torch.exp(torch.randn(L, 1, 1, generator=g) * 2.0)
        mla_ref.quantize_and_cache_k(key.to(torch.bfloat16), k_lora, k_scale, k_rope, loc)
    q = torch.randn(bs, s_q, H, 576, generator=g).to(torch.bfloat16)
    return dict(q=q, k_lora=k_lora, k_scale=k_scale, k_rope=k_rope, block_table=block_table,
                cache_seqlens=torch.tensor(lens, dtype=torch.int32), total_pages=total_pages)
