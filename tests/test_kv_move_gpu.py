"""MI355X: K7 KV row move over all layers' buffers (`fluent_mi355.kvmove`, csrc/kv_move.hip) through the C-ABI: bit-exact vs
the golden vectors of the reference's move_kv_cache_native and vs the oracle on pool-sized buffers."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import mla_ref
from test_oracle_golden import kv_move_buffers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_kv_move_bit_exact_vs_reference_golden():
    from fluent_mi355.kvmove import KVMoveTable

    g = load_golden("kv_move.npz")
    bufs = [b.to(DEV) for b in kv_move_buffers(g)]
    KVMoveTable(bufs).move(torch.from_numpy(g["tgt"]).to(DEV), torch.from_numpy(g["src"]).to(DEV))
    torch.cuda.synchronize()
    for got, want in zip(bufs, kv_move_buffers(g, "_out")):
        assert torch.equal(got.cpu().view(torch.uint8), want.view(torch.uint8))


@pytest.mark.parametrize("n,layers,shift", [(1, 3, 1), (256, 61, 3), (8192, 2, 64), (1000, 4, 0), (8193, 2, 5), (20000, 3, 77)])
def test_kv_move_compaction_vs_oracle(n, layers, shift):
    """A compaction like the speculative-decoding accept step: rows slide down by `shift` slots (sources and targets
    overlap; shift 0 = self-copy), per_token_head MLA buffers (512 B, 4 B, 128 B rows) and a bf16 [S, 576] cache, int32 locs."""
    from fluent_mi355.kvmove import KVMoveTable

    S = n + shift + 70
    g = torch.Generator().manual_seed(n + layers)
    bufs = []
    for _ in range(layers):
        bufs += [torch.randint(0, 256, (S, 1, 512), generator=g, dtype=torch.uint8), torch.rand(S, 1, 1, generator=g),
                 torch.randn(S, 1, 64, generator=g).to(torch.bfloat16)]
    bufs.append(torch.randn(S, 1, 576, generator=g).to(torch.bfloat16))
    perm = torch.randperm(n, generator=g)
    src = (torch.arange(n) + 7 + shift)[perm].to(torch.int32)
    tgt = (torch.arange(n) + 7)[perm].to(torch.int32)
    dev_bufs = [b.to(DEV) for b in bufs]
    KVMoveTable(dev_bufs).move(tgt.to(DEV), src.to(DEV))
    torch.cuda.synchronize()
    mla_ref.move_kv_cache(bufs, tgt, src)
    for got, want in zip(dev_bufs, bufs):
        assert torch.equal(got.cpu().view(torch.uint8), want.view(torch.uint8))


def test_kv_move_out_of_range_rows_are_skipped_and_limits():
    from fluent_mi355.kvmove import KVMoveTable, move_kv_cache

    buf = torch.arange(40 * 16, dtype=torch.float32).view(40, 16).to(DEV)
    before = buf.clone()
    move_kv_cache([buf], torch.tensor([3, 40, -1, 5], device=DEV), torch.tensor([4, 2, 6, 99], device=DEV))
    torch.cuda.synchronize()
    want = before.clone()
    want[3] = before[4]
    assert torch.equal(buf, want)
    move_kv_cache([buf], torch.empty(0, dtype=torch.int64, device=DEV), torch.empty(0, dtype=torch.int64, device=DEV))
    # more than 8192 rows per call go through the staged path (gather, then scatter): same skip rule, same semantics
    big = torch.arange(9000 * 16, dtype=torch.float32).view(9000, 16).to(DEV)
    before = big.clone()
    tgt = torch.arange(8500, device=DEV)
    src = torch.arange(8500, device=DEV) + 100
    tgt[17], src[18] = -1, 9000                       # two skipped pairs
    KVMoveTable([big]).move(tgt, src)
    torch.cuda.synchronize()
    want = before.clone()
    keep = torch.ones(8500, dtype=torch.bool, device=DEV)
    keep[17] = keep[18] = False
    want[tgt[keep]] = before[src[keep]]
    assert torch.equal(big, want)
    with pytest.raises(RuntimeError):
        KVMoveTable([buf, torch.zeros(41, 16, device=DEV)])
