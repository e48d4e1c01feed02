"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Everything goes through the drop-in modules
(`flash_mla_fp8`) -> ctypes -> C-ABI -> HIP kernels and is compared with the oracle / golden vectors."""
import os
import numpy as np
import pytest
import torch

from helpers import bf16_from_u16, load_golden, make_paged_case, rel_mae
from oracle import mla_ref

pytestmark = pytest.mark.gpu
SCALE = 192 ** -0.5


@pytest.fixture(scope="module")
def fm():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import flash_mla_fp8

    return flash_mla_fp8


def dev():
    return torch.device("cuda:0")


# ---------------------------------------------------------------- K5 / K6 / K4: bit-exact byte work
def test_quantize_and_cache_k_bit_exact_vs_reference_golden(fm):
    g = load_golden("kv_quant_per_token.npz")
    key = bf16_from_u16(g["key"]).to(dev())
    loc = torch.from_numpy(g["loc"]).to(dev())
    k_lora = torch.zeros(g["k_lora"].shape, dtype=torch.uint8, device=dev())
    k_scale = torch.zeros(g["k_scale"].shape, dtype=torch.float32, device=dev())
    k_rope = torch.zeros(g["k_rope"].shape, dtype=torch.bfloat16, device=dev())
    fm.quantize_and_cache_k(key=key.contiguous(), k_lora_cache=k_lora, k_lora_scale_cache=k_scale,
                            k_rope_cache=k_rope, indices=loc.to(torch.int32), head_dim_v=512)
    torch.cuda.synchronize()
    assert np.array_equal(k_lora.cpu().numpy(), g["k_lora"])          # untouched slots stay zero too
    assert np.array_equal(k_scale.cpu().numpy().view(np.uint32), g["k_scale"].view(np.uint32))
    assert np.array_equal(k_rope.cpu().view(torch.int16).numpy().view(np.uint16), g["k_rope"])


def test_dequantize_gather_bit_exact_vs_reference_golden(fm):
    g = load_golden("kv_quant_per_token.npz")
    lora, rope = fm.dequantize_ckv_fused_indexed(torch.from_numpy(g["k_lora"]).to(dev()).view(torch.float8_e4m3fn),
                                                 bf16_from_u16(g["k_rope"]).to(dev()),
                                                 torch.from_numpy(g["k_scale"]).to(dev()),
                                                 torch.from_numpy(g["gather"]).to(dev()))
    assert torch.equal(lora.cpu().view(torch.int16), bf16_from_u16(g["lora_deq"]).view(torch.int16))
    assert torch.equal(rope.cpu().view(torch.int16), bf16_from_u16(g["rope_deq"]).view(torch.int16))


def test_quant_roundtrip_random_bit_exact_vs_oracle(fm):
    g = torch.Generator().manual_seed(3)
    n, slots = 4097, 8192
    key = (torch.randn(n, 1, 576, generator=g) * torch.exp(torch.randn(n, 1, 1, generator=g) * 3)).to(torch.bfloat16)
    loc = torch.randperm(slots, generator=g)[:n].to(torch.int32)
    ref = [torch.zeros(slots, 1, 512, dtype=torch.uint8), torch.zeros(slots, 1, 1), torch.zeros(slots, 1, 64, dtype=torch.bfloat16)]
    mla_ref.quantize_and_cache_k(key, ref[0], ref[1], ref[2], loc)
    out = [torch.zeros_like(t, device=dev()) for t in ref]
    fm.quantize_and_cache_k(key.to(dev()), out[0], out[1], out[2], loc.to(dev()), 512)
    assert torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[1].cpu(), ref[1])
    assert torch.equal(out[2].cpu().view(torch.int16), ref[2].view(torch.int16))
    # Q-side (K4) same arithmetic per (token, head)
    q = key[:300].view(3, 1, 100, 576).contiguous()
    qn, qs, qr = fm.quantize_ckv_per_token_head(q.to(dev()), 512)
    rn, rs, rr = mla_ref.quantize_ckv_per_token_head(q, 512)
    assert torch.equal(qn.cpu().view(torch.uint8), rn.view(torch.uint8)) and torch.equal(qs.cpu(), rs)
    assert torch.equal(qr.cpu().view(torch.int16), rr.view(torch.int16))


def test_fused_quant_q_and_cache_k_matches_the_two_separate_calls(fm):
    """K5 + K4 in one launch (flash_mla_fp8.quantize_q_and_cache_k): bytes identical to quantize_and_cache_k followed by
    quantize_ckv_per_token_head, at decode sizes (one row per wave) and above 8192 rows (16 lanes per row, 4 rows per wave; an odd
    number of K rows so that a wave straddles the K / Q boundary); out-of-pool and negative cache locations are skipped as in K5."""
    g = torch.Generator().manual_seed(33)
    for bs, H, s_q in ((3, 16, 1), (128, 128, 1), (65, 128, 4)):
        slots = 4096
        key = (torch.randn(bs, 1, 576, generator=g) * torch.exp(torch.randn(bs, 1, 1, generator=g))).to(torch.bfloat16).to(dev())
        q = torch.randn(bs, s_q, H, 576, generator=g).to(torch.bfloat16).to(dev())
        loc = torch.randperm(slots, generator=g)[:bs].to(torch.int32)
        loc[0] = -1
        if bs > 2:
            loc[2] = slots + 5
        loc = loc.to(dev())
        a = [torch.zeros(slots, 1, 512, dtype=torch.uint8, device=dev()), torch.zeros(slots, 1, 1, device=dev()),
             torch.zeros(slots, 1, 64, dtype=torch.bfloat16, device=dev())]
        b = [torch.zeros_like(t) for t in a]
        fm.quantize_and_cache_k(key, a[0], a[1], a[2], loc, 512)
        rn, rs, rr = fm.quantize_ckv_per_token_head(q, 512)
        qn, qs, qr = fm.quantize_q_and_cache_k(q, key, b[0], b[1], b[2], loc, 512)
        torch.cuda.synchronize()
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
        assert torch.equal(qn.view(torch.uint8), rn.view(torch.uint8)) and torch.equal(qs, rs)
        assert torch.equal(qr.view(torch.int16), rr.view(torch.int16))


@pytest.mark.parametrize("H", [128, 16])
def test_decode_with_an_expanded_or_sliced_block_table(fm, H):
    """The row stride of block_table is not its row width: (a) one table row shared by every request through
    expand() (stride(0) == 0) and (b) a column slice of a wider table (stride(0) > shape[1]) give the bytes of the same launch
    on a contiguous copy of the table — for the > 32-row kernel (unconditional, clamped page-id loads) and the <= 32-row one."""
    g = torch.Generator().manual_seed(4242 + H)
    bs, L = 3, 700
    npg = (L + 63) // 64
    total = 2 * npg + 1
    key = torch.randn(total * 64, 1, 576, generator=g).to(torch.bfloat16).to(dev())
    k_lora = torch.zeros(total * 64, 1, 512, dtype=torch.uint8, device=dev())
    k_scale = torch.zeros(total * 64, 1, 1, device=dev())
    k_rope = torch.zeros(total * 64, 1, 64, dtype=torch.bfloat16, device=dev())
    fm.quantize_and_cache_k(key, k_lora, k_scale, k_rope, torch.arange(total * 64, dtype=torch.int32, device=dev()), 512)
    cache = (k_lora.view(total, 64, 1, 512), k_rope.view(total, 64, 1, 64), k_scale.view(total, 64, 1, 1))
    q = torch.randn(bs, 1, H, 576, generator=g).to(torch.bfloat16).to(dev())
    qn, qs, qr = fm.quantize_ckv_per_token_head(q, 512)
    seq = torch.full((bs,), L, dtype=torch.int32, device=dev())
    meta, ns = fm.get_mla_metadata(seq, H, 1)
    row = (torch.randperm(total - 1, generator=g)[:npg] + 1).to(torch.int32).to(dev())

    def run(bt):
        o, lse = fm.flash_mla_ckv_fp8_per_token(qn, qr, cache[0], cache[1], qs, cache[2], bt, seq, 512, meta, ns, SCALE, True)
        torch.cuda.synchronize()
        return o, lse

    shared = row.view(1, npg).expand(bs, npg)
    assert shared.stride(0) == 0
    o_ref, l_ref = run(shared.contiguous())
    o, lse = run(shared)
    assert torch.equal(o.view(torch.int16), o_ref.view(torch.int16)) and torch.equal(lse, l_ref)
    wide = torch.full((bs, npg + 37), 10 ** 6, dtype=torch.int32, device=dev())    # (page ids beyond the pool in the columns that must not be read)
    wide[:, :npg] = row
    sliced = wide[:, :npg]
    assert sliced.stride(0) == npg + 37
    o, lse = run(sliced)
    assert torch.equal(o.view(torch.int16), o_ref.view(torch.int16)) and torch.equal(lse, l_ref)


@pytest.mark.parametrize("bs,H,s_q,lens", [(5, 128, 1, [4096, 1, 63, 700, 129]), (3, 40, 1, [300, 64, 5000]), (4, 64, 4, [900, 130, 4, 2048]),
                                           (128, 128, 1, None)])
def test_decode_with_the_query_quantised_in_its_prologue_is_bit_identical(fm, bs, H, s_q, lens):
    """flash_mla_ckv_fp8_per_token_bf16_q (K4 inside K1's request prologue) == quantize_ckv_per_token_head + flash_mla_ckv_fp8_per_token:
    the same Q bytes reach the same kernel body, so outputs and LSEs are identical bit for bit — also with split requests (small
    batches on 128 parts), padded head counts (H = 40) and the causal s_q = 4 verify step."""
    g = torch.Generator().manual_seed(1000 + bs + H)
    if lens is None:
        lens = [2048 + int(torch.randint(0, 4096, (1,), generator=g)) for _ in range(bs)]
    pages_per = [(l + 63) // 64 for l in lens]
    total = sum(pages_per) + 1
    perm = (torch.randperm(total - 1, generator=g) + 1).tolist()
    bt = torch.zeros(bs, max(pages_per), dtype=torch.int32)
    k = 0
    for b, n in enumerate(pages_per):
        bt[b, :n] = torch.tensor(perm[k:k + n], dtype=torch.int32)
        k += n
    key = (torch.randn(total * 64, 1, 576, generator=g) * torch.exp(torch.randn(total * 64, 1, 1, generator=g) * 0.5)).to(torch.bfloat16).to(dev())
    k_lora = torch.zeros(total * 64, 1, 512, dtype=torch.uint8, device=dev())
    k_scale = torch.zeros(total * 64, 1, 1, device=dev())
    k_rope = torch.zeros(total * 64, 1, 64, dtype=torch.bfloat16, device=dev())
    fm.quantize_and_cache_k(key, k_lora, k_scale, k_rope, torch.arange(total * 64, dtype=torch.int32, device=dev()), 512)
    q = (torch.randn(bs, s_q, H, 576, generator=g) * torch.exp(torch.randn(bs, s_q, H, 1, generator=g))).to(torch.bfloat16).to(dev())
    seq = torch.tensor(lens, dtype=torch.int32, device=dev())
    meta, ns = fm.get_mla_metadata(seq, s_q * H, 1)
    cache = (k_lora.view(total, 64, 1, 512), k_rope.view(total, 64, 1, 64), k_scale.view(total, 64, 1, 1))
    qn, qs, qr = fm.quantize_ckv_per_token_head(q, 512)
    o1, l1 = fm.flash_mla_ckv_fp8_per_token(qn, qr, cache[0], cache[1], qs, cache[2], bt.to(dev()), seq, 512, meta, ns, SCALE, True)
    o2, l2 = fm.flash_mla_ckv_fp8_per_token_bf16_q(q, cache[0], cache[1], cache[2], bt.to(dev()), seq, 512, meta, ns, SCALE, True)
    torch.cuda.synchronize()
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    assert torch.equal(l1, l2)
    assert int(meta[:, 5:].abs().sum()) == 0        # the in-kernel merge counters are back to zero


# ---------------------------------------------------------------- K3: scheduler, bit-exact vs its Python statement
@pytest.mark.parametrize("lens,rows", [([4096] * 128, 128), ([1] * 160, 16), ([0, 5, 200, 0, 9000], 128),
                                       ([16384], 64), ([63, 64, 65, 4095, 4097] * 7, 512), ([4096] * 128, 16), ([700] * 1000, 16),
                                       ([0] * 40 + [130] * 3 + [0] * 9, 128), ([8192] * 256, 16), ([1], 128), ([65536], 16)])
def test_get_mla_metadata_bit_exact(fm, lens, rows):
    seq = torch.tensor(lens, dtype=torch.int32, device=dev())
    meta, ns = fm.get_mla_metadata(seq, rows, 1)
    rmeta, rns = mla_ref.get_mla_metadata(lens, meta.shape[0])
    assert meta.dtype == torch.int32 and ns.shape == (len(lens) + 1,)
    assert np.array_equal(meta.cpu().numpy(), rmeta) and np.array_equal(ns.cpu().numpy(), rns)
    # static in bs: same number of parts for any batch
    meta2, _ = fm.get_mla_metadata(torch.ones(7, dtype=torch.int32, device=dev()), rows, 1)
    assert meta2.shape == meta.shape


@pytest.mark.gpu
def test_get_mla_metadata_bit_exact_on_random_batches(fm):
    """K3's one-walk kernel against the Python statement on 300 seeded random batches (uniform / ragged / zero-length runs / more requests than parts /
    one long request), every rows-per-request class of the chip (1, 2, 4, 8 row groups -> 256 .. 32 parts)."""
    import random
    rnd = random.Random(20260930)
    for it in range(300):
        bs = rnd.choice([1, 2, 3, 5, 8, 16, 33, 64, 128, 200, 256, 700])
        mx = rnd.choice([1, 63, 64, 65, 500, 4096, 9000, 40000])
        kind = rnd.random()
        if kind < 0.25:
            lens = [mx] * bs
        elif kind < 0.5:
            lens = [rnd.choice([0, 1, mx]) for _ in range(bs)]
        else:
            lens = [rnd.randint(0, mx) for _ in range(bs)]
        rows = rnd.choice([16, 64, 128, 256, 512])
        seq = torch.tensor(lens, dtype=torch.int32, device=dev())
        meta, ns = fm.get_mla_metadata(seq, rows, 1)
        rmeta, rns = mla_ref.get_mla_metadata(lens, meta.shape[0])
        assert np.array_equal(meta.cpu().numpy(), rmeta) and np.array_equal(ns.cpu().numpy(), rns), (it, bs, mx, rows, lens[:8])


# ---------------------------------------------------------------- K1: decode parity
LAST_CASE = {}   # inputs of the most recent run_decode: a failing check() dumps them (a flake must leave evidence)


def _dump_failure(tag, **arrays):
    """inputs + outputs of a failed comparison -> gpurun_out/failures/<tag>.npz (merged back from the GPU box)"""
    import os
    import numpy as np
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "failures")
    os.makedirs(root, exist_ok=True)
    out = {}
    for k, v in {**LAST_CASE, **arrays}.items():
        if torch.is_tensor(v) and v.numel() * v.element_size() <= (64 << 20):
            t = v.detach().cpu()
            out[k] = (t.view(torch.int16) if t.dtype == torch.bfloat16 else t.view(torch.uint8) if t.dtype == torch.float8_e4m3fn else t).numpy()
        elif isinstance(v, (int, float, bool, str)):
            out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(root, "".join(ch if ch.isalnum() or ch in "-_" else "_" for ch in str(tag)) + ".npz"), **out)


def run_decode(fm, c, H, s_q=1, causal=True, emulate=True):
    LAST_CASE.clear()
    LAST_CASE.update({k: v for k, v in c.items()}, H=H, s_q=s_q, causal=causal)
    d = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in c.items()}
    pages = c["total_pages"]
    qn, qs, qr = fm.quantize_ckv_per_token_head(d["q"].contiguous(), 512)
    meta, ns = fm.get_mla_metadata(d["cache_seqlens"], s_q * H, 1)
    o, lse = fm.flash_mla_ckv_fp8_per_token(
        q_nope=qn, q_rope=qr, k_cache_lora=d["k_lora"].view(pages, 64, 1, 512),
        k_cache_rope=d["k_rope"].view(pages, 64, 1, 64), q_scale=qs, k_scale=d["k_scale"].view(pages, 64, 1, 1),
        block_table=d["block_table"], cache_seqlens=d["cache_seqlens"], head_dim_v=512,
        tile_scheduler_metadata=meta, num_splits=ns, softmax_scale=SCALE, causal=causal)
    torch.cuda.synchronize()
    args = (qn.cpu(), qs.cpu(), qr.cpu(), c["k_lora"].view(pages, 64, 1, 512), c["k_scale"].view(pages, 64, 1, 1),
            c["k_rope"].view(pages, 64, 1, 64), c["block_table"], c["cache_seqlens"], SCALE, causal)
    ref, rlse = mla_ref.mla_decode_fp8_per_token(*args)
    if emulate:
        emu, _ = mla_ref.mla_decode_fp8_per_token_emulated(*args)
        e = (o.cpu().double() - emu).abs()
        r = float(e.mean() / emu.abs().mean().clamp_min(1e-30))
        # the kernel must do exactly what its design says (fp32-vs-fp64 accumulation and rare 1-ulp P' flips only)
        assert r < 3e-3, ("vs bit-level statement of the kernel", r)
    return o.cpu(), lse.cpu(), ref, rlse, ns.cpu()


K1_REL_MAE_BOUND = 3.2e-2   # = 1.15 x 2.74e-2, the maximum over 11 shapes x 200 seeds (profiles/r03_mla_seed_sweep.json)

MEASURED = []   # (tag, rel-MAE, max-abs, max|ref|, max LSE error): written out by test_zz_write_measured_errors


def check(o, lse, ref, rlse, tag):
    try:
        return _check(o, lse, ref, rlse, tag)
    except AssertionError:
        _dump_failure(tag, o=o, lse=lse, ref=ref, rlse=rlse)
        raise


def _check(o, lse, ref, rlse, tag):
    assert torch.isfinite(o.float()).all(), tag
    err = (o.double() - ref).abs()
    rel = float(err.mean() / ref.abs().mean().clamp_min(1e-30))
    _fin = torch.isfinite(rlse)
    MEASURED.append({"case": str(tag), "rel_mae": rel, "max_abs": float(err.max()), "max_ref": float(ref.abs().max()),
                     "lse_err": float((lse.double()[_fin] - rlse[_fin]).abs().max()) if bool(_fin.any()) else 0.0})
    # stated FP8 tolerance vs dequantise-then-exact attention: rel-MAE < K1_REL_MAE_BOUND = 1.15 x the WORST case of a seed
    # sweep (tools/seed_sweep_mla.py: every CASES shape x 200 seeds -> profiles/r03_mla_seed_sweep.json; cfg1 — seq 128,
    # 16 heads, the smallest sample — has the widest spread: 1.4e-2 .. 2.74e-2, mean 2.1e-2), max-abs < 1e-1 on N(0,1)-scaled
    # data (|o| <= ~4; for larger outputs the absolute bound scales with the data).  SURVEY section 8c's 2e-2 is what an
    # exact P would give; here — as in the reference's own flashmla-fp8, whose PV GEMM is fp8 x fp8 — P is re-quantised to
    # e4m3 for the MX PV MFMA (3 mantissa bits, rms relative rounding 2^-4/sqrt(3) = 3.6 % per weight): on i.i.d. V rows
    # signal and rounding noise both scale as 1/sqrt(N_eff), so the mean sits at 1.2-2.2e-2 for every length (the CPU
    # emulation of the kernel's arithmetic reproduces the distribution: same tool with --emulated).
    # The bit-level statement of the kernel is checked to 3e-3 in run_decode.
    assert rel < K1_REL_MAE_BOUND, (tag, rel)
    # max-abs: 1e-1 on N(0,1) data for ordinary lengths; 2-3-token sequences are the worst case of fp8 weights: the weight
    # ratio moves by <= 2*2^-4*w1*w2 <= 3.1e-2, times |v1-v2| <= 2 max|v|  ->  bound 5e-2 * max|o| covers it
    assert float(err.max()) < max(1e-1, 5e-2 * float(ref.abs().max())), (tag, float(err.max()))
    fin = torch.isfinite(rlse)
    assert torch.equal(torch.isfinite(lse), fin), tag
    lerr = (lse.double()[fin] - rlse[fin]).abs()
    assert bool((lerr < 2e-2 + 1e-4 * rlse[fin].abs()).all()), (tag, float(lerr.max()))   # fp32 score arithmetic
    return rel


CASES = [
    ("cfg1", [128], 16, 1),
    ("ragged_pageedges", [1, 63, 64, 65, 130, 200], 16, 1),
    ("h128", [200, 77, 1000], 128, 1),
    ("h64_rowgroup", [333, 64], 64, 1),
    ("h40_padrows", [257], 40, 1),
    ("empty_and_one", [0, 1, 0, 2], 16, 1),
    ("split_long", [9000], 128, 1),
    ("split_mixed", [5000, 3, 700, 2500], 32, 1),
    ("mtp_sq4", [68, 4, 300], 16, 4),
    ("mtp_sq4_h128", [260, 129], 128, 4),
    ("sq2_h8", [5, 64, 66], 8, 2),
]


CASE_SEEDS = {c[0]: 100 + i for i, c in enumerate(CASES)}


@pytest.mark.parametrize("name,lens,H,s_q", CASES, ids=[c[0] for c in CASES])
def test_decode_parity_vs_oracle(fm, name, lens, H, s_q):
    # fixed seed per case (NOT hash(name): Python string hashes are randomised per process, so every run drew new inputs)
    c = make_paged_case(lens, H, s_q=s_q, seed=CASE_SEEDS[name])
    o, lse, ref, rlse, ns = run_decode(fm, c, H, s_q)
    check(o, lse, ref, rlse, name)


def test_decode_wide_kscale_spread_and_spikes(fm):
    """per-token scales spread over e^±6 and one dominant key per row (forces the defer-max rescale branch)."""
    c = make_paged_case([700, 130], 128, seed=11, kscale_spread=True)
    o, lse, ref, rlse, _ = run_decode(fm, c, 128)
    check(o, lse, ref, rlse, "kscale_spread")


def test_decode_rescale_branch_spike(fm):
    from oracle import mla_ref as R

    c = make_paged_case([640], 64, seed=12)
    # plant a key late in the sequence that matches query row 3 strongly -> running max jumps by >> 2^4 at page 8
    t = 600
    slot = int(c["block_table"][0, t // 64]) * 64 + t % 64
    key = torch.zeros(1, 1, 576)
    key[0, 0, :512] = c["q"][0, 0, 3, :512].float() * 1.5
    R.quantize_and_cache_k(key.to(torch.bfloat16), c["k_lora"], c["k_scale"], c["k_rope"], torch.tensor([slot], dtype=torch.int32))
    o, lse, ref, rlse, _ = run_decode(fm, c, 64)
    check(o, lse, ref, rlse, "spike")


@pytest.mark.parametrize("H", [16, 64, 128])
def test_decode_reference_jump_beyond_block_scale_range(fm, H):
    """A key late in the sequence beats the first page's maximum by ~148 nats (213 log2 units) for one query row: beyond the 128 log2 units
    of headroom the O reference starts with, so the kernel moves that row's reference up in place (O *= 2^-k in the step's out-of-line block;
    rounds 2-4 repeated the request — the `redo` pass).  Every row count runs the same kernel template: 4 waves at H = 16, 8 at H = 64 / 128."""
    from oracle import mla_ref as R

    c = make_paged_case([1500, 200], H, seed=21)
    t = 1400                                                 # page 21 of request 0 (second split part at H = 128)
    slot = int(c["block_table"][0, t // 64]) * 64 + t % 64
    key = torch.zeros(1, 1, 576)
    key[0, 0, :512] = c["q"][0, 0, 3, :512].float() * 4.0    # logit ~ 4 |q|^2 / sqrt(192) >> 69 nats above the rest
    R.quantize_and_cache_k(key.to(torch.bfloat16), c["k_lora"], c["k_scale"], c["k_rope"], torch.tensor([slot], dtype=torch.int32))
    o, lse, ref, rlse, _ = run_decode(fm, c, H)
    check(o, lse, ref, rlse, f"jump_H{H}")
    # the peaked row returns (almost exactly) the planted token's latent
    assert float((o[0, 0, 3].double() - ref[0, 0, 3]).abs().max()) < 5e-2 * float(ref[0, 0, 3].abs().max())


def test_decode_vs_reference_backend_golden(fm):
    """End to end against the REAL reference's TorchNativeAttnBackend output (bf16 KV): quantise the golden cache
    with K5, quantise q with K4, decode with K1.  Tolerance covers per-token FP8 of K and q."""
    for name in ("cfg1", "ragged", "h128"):
        g = load_golden(f"mla_torch_native_{name}.npz")
        H = int(g["H"])
        kv = bf16_from_u16(g["kv_buffer_after"]).to(dev())          # [slots,1,576]
        slots = kv.shape[0]
        k_lora = torch.zeros(slots, 1, 512, dtype=torch.uint8, device=dev())
        k_scale = torch.ones(slots, 1, 1, dtype=torch.float32, device=dev())
        k_rope = torch.zeros(slots, 1, 64, dtype=torch.bfloat16, device=dev())
        fm.quantize_and_cache_k(kv.contiguous(), k_lora, k_scale, k_rope,
                                torch.arange(slots, dtype=torch.int32, device=dev()), 512)
        q = bf16_from_u16(g["q"]).to(dev()).view(-1, 1, H, 576)
        seq = torch.from_numpy(g["seq_lens"]).to(dev())
        bt = torch.from_numpy(g["block_table"]).to(dev())
        qn, qs, qr = fm.quantize_ckv_per_token_head(q.contiguous(), 512)
        meta, ns = fm.get_mla_metadata(seq, H, 1)
        pages = slots // 64
        o, _ = fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                              k_scale.view(pages, 64, 1, 1), bt, seq, 512, meta, ns, float(g["scaling"]), True)
        ref = bf16_from_u16(g["o"]).view(-1, 1, H, 512)
        r = rel_mae(o.cpu(), ref)
        assert r < 6e-2, (name, r)   # fp8 K (3 mantissa bits) + fp8 q vs the bf16 reference
        # a 1-token sequence returns the dequantised latent itself: per-token e4m3 step near amax~4.3 is 0.3 -> 0.16
        assert float((o.cpu().float() - ref.float()).abs().max()) < 2e-1, name


def _full_size_properties(fm, lens, H, samples, tag, s_q=1):
    """Full-size checks where the oracle is too slow for the whole batch — size-independent properties:
    (1) o is a convex combination of V rows: |o| <= max|V|; (2) moving every page to another physical location (same
    logical content) leaves the output BIT-identical; (3) a sampled subset of requests matches the oracle."""
    bs = len(lens)
    npgs = [(L + 63) // 64 for L in lens]
    mp, pages = max(npgs), sum(npgs) + 1
    g = torch.Generator(device="cuda").manual_seed(len(lens) + H)
    key = torch.randn(pages * 64, 1, 576, device=dev(), generator=g, dtype=torch.float32).to(torch.bfloat16)
    k_lora = torch.empty(pages * 64, 1, 512, dtype=torch.uint8, device=dev())
    k_scale = torch.empty(pages * 64, 1, 1, dtype=torch.float32, device=dev())
    k_rope = torch.empty(pages * 64, 1, 64, dtype=torch.bfloat16, device=dev())
    fm.quantize_and_cache_k(key, k_lora, k_scale, k_rope, torch.arange(pages * 64, dtype=torch.int32, device=dev()), 512)
    del key
    perm = (torch.randperm(pages - 1, device=dev(), generator=g).to(torch.int32) + 1).cpu()
    bt = torch.zeros(bs, mp, dtype=torch.int32)          # unused tail entries point at the padding page 0
    o0 = 0
    for b, n in enumerate(npgs):
        bt[b, :n] = perm[o0:o0 + n]
        o0 += n
    bt = bt.to(dev())
    seq = torch.tensor(lens, dtype=torch.int32, device=dev())
    q = torch.randn(bs, s_q, H, 576, device=dev(), generator=g, dtype=torch.float32).to(torch.bfloat16)
    qn, qs, qr = fm.quantize_ckv_per_token_head(q, 512)
    meta, ns = fm.get_mla_metadata(seq, s_q * H, 1)

    def run(kl, ks, kr, table):
        return fm.flash_mla_ckv_fp8_per_token(qn, qr, kl.view(pages, 64, 1, 512), kr.view(pages, 64, 1, 64), qs,
                                              ks.view(pages, 64, 1, 1), table, seq, 512, meta, ns, SCALE, True)

    o, lse = run(k_lora, k_scale, k_rope, bt)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all(), tag
    vmax = (k_lora.view(torch.float8_e4m3fn).float() * k_scale).abs().amax()
    assert float(o.float().abs().max()) <= float(vmax) * 1.01, tag
    # (2) physical page permutation invariance (bit-exact)
    perm2 = torch.randperm(pages - 1, device=dev(), generator=g) + 1
    inv = torch.zeros(pages, dtype=torch.long, device=dev())
    inv[1:] = perm2                                           # old page p -> new page inv[p]; page 0 stays
    kl2, ks2, kr2 = torch.empty_like(k_lora), torch.empty_like(k_scale), torch.empty_like(k_rope)
    for src, dst in ((k_lora, kl2), (k_scale, ks2), (k_rope, kr2)):
        dst.view(pages, -1)[inv] = src.view(pages, -1)
    o2, lse2 = run(kl2, ks2, kr2, inv[bt.long()].to(torch.int32).contiguous())
    assert torch.equal(o.view(torch.int16), o2.view(torch.int16)) and torch.equal(lse, lse2), tag
    del kl2, ks2, kr2
    # (3) sampled requests vs oracle
    kl_c, ks_c, kr_c = k_lora.cpu().view(pages, 64, 1, 512), k_scale.cpu().view(pages, 64, 1, 1), k_rope.cpu().view(pages, 64, 1, 64)
    for b in samples:
        ref, rlse = mla_ref.mla_decode_fp8_per_token(qn[b:b + 1].cpu(), qs[b:b + 1].cpu(), qr[b:b + 1].cpu(), kl_c, ks_c, kr_c,
                                                     bt[b:b + 1].cpu(), seq[b:b + 1].cpu(), SCALE, True)
        check(o[b:b + 1].cpu(), lse[b:b + 1].cpu(), ref, rlse, f"{tag} req {b} (len {lens[b]})")


def test_full_size_properties_cfg5_mtp_verify(fm):
    """BASELINE config 5 at full size: MTP verify, s_q=4 (causal over the 4 draft tokens), bs=64, seq=16384, H=64 — 256 query
    rows per request; cache_seqlens include the s_q new tokens (flashmla_backend.py:135-136)."""
    _full_size_properties(fm, [16384] * 64, 64, (0, 63), "cfg5", s_q=4)


def test_full_size_properties_bs128_seq4096(fm):
    """BASELINE config 2 at full size (bs=128, seq=4096, H=128)."""
    _full_size_properties(fm, [4096] * 128, 128, (0, 77, 127), "cfg2")


def test_full_size_properties_cfg2_ragged(fm):
    """BASELINE config 2, ragged variant: lengths uniform in 2048..6144 (mean 4096), page edges included."""
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(2048, 6145, (128,), generator=g).tolist()
    lens[3], lens[90] = 2048, 6144
    lens[17] = 4097                      # one token on its last page
    _full_size_properties(fm, lens, 128, (3, 17, 90, 127), "cfg2-ragged")


def test_full_size_properties_cfg4_shape(fm):
    """BASELINE config 4's attention shape on one rank of attention-TP8: 16 heads per GPU, bs=256, seq=8192 (every rank
    streams all requests' latent KV)."""
    _full_size_properties(fm, [8192] * 256, 16, (0, 131, 255), "cfg4-H16")


def test_graph_capture_replay(fm):
    """The decode path must be hipGraph-capturable with static buffers (cuda_graph_runner.py:433-434, flashmla_backend.py:366-405)."""
    c = make_paged_case([300, 64, 129], 16, seed=5)
    d = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in c.items()}
    pages = c["total_pages"]
    meta, ns = fm.get_mla_metadata(d["cache_seqlens"], 16, 1)
    q_static = d["q"].clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            qn, qs, qr = fm.quantize_ckv_per_token_head(q_static, 512)
            fm.flash_mla_ckv_fp8_per_token(qn, qr, d["k_lora"].view(pages, 64, 1, 512), d["k_rope"].view(pages, 64, 1, 64), qs,
                                           d["k_scale"].view(pages, 64, 1, 1), d["block_table"], d["cache_seqlens"], 512, meta, ns, SCALE, True)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        qn, qs, qr = fm.quantize_ckv_per_token_head(q_static, 512)
        o, lse = fm.flash_mla_ckv_fp8_per_token(qn, qr, d["k_lora"].view(pages, 64, 1, 512), d["k_rope"].view(pages, 64, 1, 64), qs,
                                                d["k_scale"].view(pages, 64, 1, 1), d["block_table"], d["cache_seqlens"], 512, meta, ns, SCALE, True)
    q_static.copy_(torch.randn_like(q_static))
    graph.replay()
    torch.cuda.synchronize()
    c["q"] = q_static.cpu()
    o_ref, lse_e, ref, rlse, _ = run_decode(fm, c, 16)
    assert torch.equal(o.cpu().view(torch.int16), o_ref.view(torch.int16))


@pytest.mark.parametrize("H", [16, 128])
def test_mtp_verify_and_draft_steps_in_one_graph(fm, H):
    """BASELINE config 5 in miniature: one speculative-decode step — verify (s_q = 4 draft tokens, causal inside the
    block, cache_seqlens = seq_lens + 4: flashmla_backend.py:105-176) followed by 3 draft decode steps whose backends
    see seq_lens + (i + 1) more tokens (FlashMLAMultiStepDecodeBackend, :411-477) — captured ONCE in a hipGraph over
    persistent buffers and replayed for two different batches, like CudaGraphRunner.replay (cuda_graph_runner.py:460-525):
    the batch is padded to the captured size with requests of seq_len fill value 1 whose block-table row points at
    padding page 0; the scheduler metadata of every backend is recomputed OUTSIDE the graph and copied into the captured
    buffers (flashmla_backend.py:366-405).  K5 (store the new tokens' KV) and K4 (quantise q) run inside the graph."""
    from oracle import mla_ref as R

    DRAFT, STEPS, BS_CAP = 4, 3, 4
    dv = dev()

    def batch(lens, seed):
        c = make_paged_case([L + DRAFT + STEPS for L in lens], H, seed=seed)      # pages for every token the step adds
        g = torch.Generator().manual_seed(seed + 1)
        nb = len(lens)
        c["q_verify"] = torch.randn(nb, DRAFT, H, 576, generator=g).to(torch.bfloat16)
        c["q_draft"] = [torch.randn(nb, 1, H, 576, generator=g).to(torch.bfloat16) for _ in range(STEPS)]
        c["k_verify"] = torch.randn(nb * DRAFT, 1, 576, generator=g).to(torch.bfloat16)
        c["k_draft"] = [torch.randn(nb, 1, 576, generator=g).to(torch.bfloat16) for _ in range(STEPS)]
        c["lens"] = lens
        return c

    cases = [batch([300, 1000, 77], 31), batch([64, 5], 32)]
    pool_pages = max(c["total_pages"] for c in cases)
    max_pages = max(c["block_table"].shape[1] for c in cases)
    slots = pool_pages * 64
    # ---- persistent graph buffers (flashmla_backend.py:289-322; kv indices filled with 1, seq_len fill value 1) ----
    k_lora = torch.zeros(slots, 1, 512, dtype=torch.uint8, device=dv)
    k_scale = torch.ones(slots, 1, 1, dtype=torch.float32, device=dv)
    k_rope = torch.zeros(slots, 1, 64, dtype=torch.bfloat16, device=dv)
    bt = torch.ones(BS_CAP, max_pages, dtype=torch.int32, device=dv)
    seq_v = torch.ones(BS_CAP, dtype=torch.int32, device=dv)
    seq_d = [torch.ones(BS_CAP, dtype=torch.int32, device=dv) for _ in range(STEPS)]
    qv = torch.zeros(BS_CAP, DRAFT, H, 576, dtype=torch.bfloat16, device=dv)
    qd = [torch.zeros(BS_CAP, 1, H, 576, dtype=torch.bfloat16, device=dv) for _ in range(STEPS)]
    kv_new = torch.zeros(BS_CAP * DRAFT, 1, 576, dtype=torch.bfloat16, device=dv)
    kd_new = [torch.zeros(BS_CAP, 1, 576, dtype=torch.bfloat16, device=dv) for _ in range(STEPS)]
    loc_v = torch.zeros(BS_CAP * DRAFT, dtype=torch.int32, device=dv)           # padded tokens write padding page 0
    loc_d = [torch.zeros(BS_CAP, dtype=torch.int32, device=dv) for _ in range(STEPS)]
    meta_v, ns_v = fm.get_mla_metadata(seq_v, DRAFT * H, 1)
    meta_d, ns_d = zip(*[fm.get_mla_metadata(seq_d[i], H, 1) for i in range(STEPS)])
    views = dict(k_cache_lora=k_lora.view(pool_pages, 64, 1, 512), k_cache_rope=k_rope.view(pool_pages, 64, 1, 64),
                 k_scale=k_scale.view(pool_pages, 64, 1, 1))

    def step():
        outs = []
        fm.quantize_and_cache_k(kv_new, k_lora, k_scale, k_rope, loc_v, 512)
        qn, qs, qr = fm.quantize_ckv_per_token_head(qv, 512)
        outs.append(fm.flash_mla_ckv_fp8_per_token(q_nope=qn, q_rope=qr, q_scale=qs, block_table=bt, cache_seqlens=seq_v,
                                                   head_dim_v=512, tile_scheduler_metadata=meta_v, num_splits=ns_v,
                                                   softmax_scale=SCALE, causal=True, **views)[0])
        for i in range(STEPS):
            fm.quantize_and_cache_k(kd_new[i], k_lora, k_scale, k_rope, loc_d[i], 512)
            qn, qs, qr = fm.quantize_ckv_per_token_head(qd[i], 512)
            outs.append(fm.flash_mla_ckv_fp8_per_token(q_nope=qn, q_rope=qr, q_scale=qs, block_table=bt,
                                                       cache_seqlens=seq_d[i], head_dim_v=512,
                                                       tile_scheduler_metadata=meta_d[i], num_splits=ns_d[i],
                                                       softmax_scale=SCALE, causal=True, **views)[0])
        return outs

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step()

    for c in cases:
        lens, nb = c["lens"], len(c["lens"])
        P = c["total_pages"] * 64
        # ---- "replay": copy the batch into the persistent buffers, re-plan every backend, replay ----
        k_lora[:P].copy_(c["k_lora"]); k_scale[:P].copy_(c["k_scale"]); k_rope[:P].copy_(c["k_rope"])
        bt.zero_(); bt[:nb, :c["block_table"].shape[1]].copy_(c["block_table"])     # padded rows -> page 0
        L = torch.tensor(lens, dtype=torch.int32)
        seq_v.fill_(1); seq_v[:nb].copy_(L + DRAFT)
        for i in range(STEPS):
            seq_d[i].fill_(1); seq_d[i][:nb].copy_(L + DRAFT + i + 1)
        cpu_bt = c["block_table"]
        def loc_of(b, t):
            return int(cpu_bt[b, t // 64]) * 64 + t % 64
        lv = torch.zeros(BS_CAP * DRAFT, dtype=torch.int32)
        for b in range(nb):
            for j in range(DRAFT):
                lv[b * DRAFT + j] = loc_of(b, lens[b] + j)
        loc_v.copy_(lv)
        kv_new.zero_(); kv_new[:nb * DRAFT].copy_(c["k_verify"])
        qv.zero_(); qv[:nb].copy_(c["q_verify"])
        lds = []
        for i in range(STEPS):
            ld = torch.zeros(BS_CAP, dtype=torch.int32)
            for b in range(nb):
                ld[b] = loc_of(b, lens[b] + DRAFT + i)
            lds.append(ld)
            loc_d[i].copy_(ld)
            kd_new[i].zero_(); kd_new[i][:nb].copy_(c["k_draft"][i])
            qd[i].zero_(); qd[i][:nb].copy_(c["q_draft"][i])
        m, n_ = fm.get_mla_metadata(seq_v, DRAFT * H, 1)
        meta_v.copy_(m); ns_v.copy_(n_)
        for i in range(STEPS):
            m, n_ = fm.get_mla_metadata(seq_d[i], H, 1)
            meta_d[i].copy_(m); ns_d[i].copy_(n_)
        graph.replay()
        torch.cuda.synchronize()
        # ---- oracle: the same stores and attentions on the CPU copy of the cache ----
        kl, ks, kr = c["k_lora"].clone(), c["k_scale"].clone(), c["k_rope"].clone()
        pages = c["total_pages"]
        def attend(q, seqlens):
            qn, qs, qr = R.quantize_ckv_per_token_head(q, 512)
            return R.mla_decode_fp8_per_token(qn, qs, qr, kl.view(pages, 64, 1, 512), ks.view(pages, 64, 1, 1),
                                              kr.view(pages, 64, 1, 64), cpu_bt, seqlens, SCALE, True)[0]
        R.quantize_and_cache_k(c["k_verify"], kl, ks, kr, lv[:nb * DRAFT])
        refs = [attend(c["q_verify"], L + DRAFT)]
        for i in range(STEPS):
            R.quantize_and_cache_k(c["k_draft"][i], kl, ks, kr, lds[i][:nb])
            refs.append(attend(c["q_draft"][i], L + DRAFT + i + 1))
        for i, (o, ref) in enumerate(zip(outs, refs)):
            got = o[:nb].cpu().reshape(ref.shape).double()
            assert torch.isfinite(o.float()).all(), i                       # padded requests included
            rel = float((got - ref).abs().mean() / ref.abs().mean())
            assert rel < 3e-2, (H, i, rel)
        # the stores inside the graph produced the oracle's bytes (padded tokens wrote only into padding page 0)
        assert torch.equal(k_lora[64:P].cpu(), kl[64:]) and torch.equal(k_rope[64:P].cpu().view(torch.int16), kr[64:].view(torch.int16))


# ---------------------------------------------------------------- K2: single fp8 [.,576] cache, scalar descales
def make_fp8_576_case(lens, H, s_q, seed):
    g = torch.Generator().manual_seed(seed)
    bs = len(lens)
    npg = [(max(L, 0) + 63) // 64 for L in lens]
    pages = sum(npg) + 3
    kc = torch.full((pages * 64, 1, 576), 0x7F, dtype=torch.uint8)       # NaN patterns outside the valid tokens
    perm = (torch.randperm(pages - 1, generator=g) + 1).tolist()
    bt = torch.zeros(bs, max(max(npg), 1) + 1, dtype=torch.int32)
    pi = 0
    for b, L in enumerate(lens):
        for j in range(npg[b]):
            bt[b, j] = perm[pi]
            pi += 1
        if L > 0:
            t = torch.arange(L)
            loc = bt[b, (t // 64).long()].long() * 64 + t % 64
            kc[loc] = torch.randn(L, 1, 576, generator=g).to(torch.float8_e4m3fn).view(torch.uint8)
    q = torch.randn(bs, s_q, H, 576, generator=g).to(torch.float8_e4m3fn)
    return q, kc, bt, torch.tensor(lens, dtype=torch.int32), pages


@pytest.mark.parametrize("lens,H,s_q,dq,dk", [([128], 16, 1, 1.0, 1.0), ([1, 63, 65, 300], 128, 1, 1.0, 1.0),
                                               ([70, 4, 200], 16, 4, 0.5, 2.0), ([3000], 64, 1, 1.0, 1.0),
                                               # the role-specialised kernel's FMT = 1 instantiation: split requests, non-unit
                                               # descales, three row groups
                                               ([2500, 700, 64, 1300], 128, 1, 0.25, 3.0), ([500, 129], 96, 2, 1.0, 0.5)])
def test_flash_mla_with_kvcache_fp8(fm, lens, H, s_q, dq, dk):
    q, kc, bt, seq, pages = make_fp8_576_case(lens, H, s_q, seed=H + s_q)
    meta, ns = fm.get_mla_metadata(seq.to(dev()), s_q * H, 1)
    dqt, dkt = torch.tensor([dq], device=dev()), torch.tensor([dk], device=dev())
    o, lse = fm.flash_mla_with_kvcache(q=q.to(dev()), k_cache=kc.to(dev()).view(torch.float8_e4m3fn).view(pages, 64, 1, 576),
                                       block_table=bt.to(dev()), cache_seqlens=seq.to(dev()), head_dim_v=512,
                                       tile_scheduler_metadata=meta, num_splits=ns, softmax_scale=SCALE, causal=True,
                                       descale_q=dqt, descale_k=dkt)
    torch.cuda.synchronize()
    ref, rlse = mla_ref.mla_decode_with_kvcache(q, kc.view(pages, 64, 1, 576), bt, seq, 512, SCALE, True, dq, dk)
    check(o.cpu(), lse.cpu(), ref, rlse, f"fp8_576 {lens}")
    # bit-level statement of the kernel: same arithmetic with constant scales and fp8 rope.  The mapping for more than 32
    # query rows (mla_decode_fp8_y.hip, FMT = 1) keeps descale_k OUT of P' (scores carry dq*dk, the output is multiplied by
    # dk in the epilogue); the mapping for <= 32 rows folds log2(dk) into P' like the per-token format
    bsz = len(lens)
    y_map = s_q * H > 32
    emu, _ = mla_ref.mla_decode_fp8_per_token_emulated(
        q[..., :512].contiguous(), torch.full((bsz, s_q, H, 1), dq * dk if y_map else dq), q[..., 512:].float(),
        kc[..., :512].contiguous().view(pages, 64, 1, 512), torch.full((pages, 64, 1, 1), 1.0 if y_map else dk),
        kc[..., 512:].contiguous().view(torch.float8_e4m3fn).float().view(pages, 64, 1, 64), bt, seq, SCALE, True)
    if y_map:
        emu = emu * dk
    e = (o.cpu().double() - emu).abs()
    assert float(e.mean() / emu.abs().mean().clamp_min(1e-30)) < 3e-3


# ---------------------------------------------------------------------------------------------------------------------
# K2-bf16: flash_mla_swap / flash_mla_fp8 .flash_mla_with_kvcache over a bf16 [pages,64,1,576] cache
# (flashmla_backend.py:163-175, 240-254).  Oracle: exact attention in float64 over the same bf16 values; tolerance =
# bf16 rounding of P (2^-9 per weight) and of the output.
# ---------------------------------------------------------------------------------------------------------------------
def make_bf16_576_case(lens, H, s_q, seed):
    g = torch.Generator().manual_seed(seed)
    bs = len(lens)
    npg = [(L + 63) // 64 for L in lens]
    pages = sum(npg) + 2
    kc = torch.full((pages * 64, 1, 576), float("nan"), dtype=torch.bfloat16)   # NaN outside the valid tokens
    perm = (torch.randperm(pages - 1, generator=g) + 1).tolist()
    bt = torch.zeros(bs, max(max(npg), 1) + 1, dtype=torch.int32)
    pi = 0
    for b, L in enumerate(lens):
        for j in range(npg[b]):
            bt[b, j] = perm[pi]
            pi += 1
        if L > 0:
            t = torch.arange(L)
            loc = bt[b, (t // 64).long()].long() * 64 + t % 64
            kc[loc] = torch.randn(L, 1, 576, generator=g).to(torch.bfloat16)
    q = torch.randn(bs, s_q, H, 576, generator=g).to(torch.bfloat16)
    return q, kc, bt, torch.tensor(lens, dtype=torch.int32), pages


@pytest.mark.parametrize("lens,H,s_q", [([128], 16, 1), ([1, 63, 65, 300], 128, 1), ([70, 4, 200], 16, 4),
                                         ([3000], 64, 1), ([33, 0, 31, 97], 8, 2), ([5000, 777], 128, 2)])
def test_flash_mla_with_kvcache_bf16(fm, lens, H, s_q):
    import flash_mla_swap
    q, kc, bt, seq, pages = make_bf16_576_case(lens, H, s_q, seed=7 * H + s_q)
    meta, ns = flash_mla_swap.get_mla_metadata(seq.to(dev()), s_q * H, 1)
    o, lse = flash_mla_swap.flash_mla_with_kvcache(q=q.to(dev()), k_cache=kc.to(dev()).view(pages, 64, 1, 576),
                                                   block_table=bt.to(dev()), cache_seqlens=seq.to(dev()), head_dim_v=512,
                                                   tile_scheduler_metadata=meta, num_splits=ns, softmax_scale=SCALE,
                                                   causal=True)
    torch.cuda.synchronize()
    ref, rlse = mla_ref.mla_decode_with_kvcache(q, torch.nan_to_num(kc).view(pages, 64, 1, 576), bt, seq, 512, SCALE, True)
    o, lse = o.cpu(), lse.cpu()
    assert torch.isfinite(o.float()).all(), "NaN/inf leaked from outside the valid tokens"
    r = rel_mae(o, ref)
    assert r < 6e-3, (lens, r)                                   # bf16 P (2^-9) + bf16 output rounding
    assert float((o.double() - ref).abs().max()) < 5e-2, lens
    m = torch.isfinite(rlse)
    assert torch.equal(torch.isfinite(lse), m)
    assert float((lse[m].double() - rlse[m]).abs().max()) < 2e-3, lens


@pytest.mark.parametrize("H", [16, 64, 128])
def test_bf16_reference_jump_moves_the_row_reference_in_place(fm, H):
    """mla_decode_bf16.hip keeps a row's reference FIXED 24 log2 units above its first tile's maximum and moves it only when a later tile beats
    it by more than 2^64 (O *= 2^-k out of line, l alike).  A key planted late in the sequence that matches query row 3 with a logit ~190 log2
    units above the rest forces that move; the other rows of the request and the second request must not notice.  4 waves at H = 16, 8 above."""
    import flash_mla_swap
    lens = [1500, 200]
    q, kc, bt, seq, pages = make_bf16_576_case(lens, H, 1, seed=31 + H)
    t = 1400
    slot = int(bt[0, t // 64]) * 64 + t % 64
    kc[slot, 0, :512] = (q[0, 0, 3, :512].float() * 4.0).to(torch.bfloat16)
    kc[slot, 0, 512:] = 0
    meta, ns = flash_mla_swap.get_mla_metadata(seq.to(dev()), H, 1)
    o, lse = flash_mla_swap.flash_mla_with_kvcache(q.to(dev()), kc.to(dev()).view(pages, 64, 1, 576), bt.to(dev()), seq.to(dev()), 512, meta, ns,
                                                   SCALE, True)
    torch.cuda.synchronize()
    ref, rlse = mla_ref.mla_decode_with_kvcache(q, torch.nan_to_num(kc).view(pages, 64, 1, 576), bt, seq, 512, SCALE, True)
    o, lse = o.cpu(), lse.cpu()
    jump = float(rlse[0, 3, 0] - rlse[0, 2, 0]) * 1.4427
    assert jump > 100, jump                                       # (the planted logit really is beyond the 64 + 24 log2 units of slack)
    assert torch.isfinite(o.float()).all()
    assert rel_mae(o, ref) < 6e-3
    assert float((lse.double() - rlse).abs().max()) < 2e-3 * max(1.0, float(rlse.abs().max()) / 50)
    # the peaked row returns (almost exactly) the planted token's latent
    assert float((o[0, 0, 3].double() - ref[0, 0, 3]).abs().max()) < 2e-2 * float(ref[0, 0, 3].abs().max())


def test_bf16_decode_vs_reference_backend_golden(fm):
    """The golden vectors are outputs of the REAL reference's TorchNativeAttnBackend over a bf16 KV buffer: the bf16
    kernel is compared with them directly (no quantisation in between)."""
    import flash_mla_swap
    for name in ("cfg1", "ragged", "h128"):
        g = load_golden(f"mla_torch_native_{name}.npz")
        H = int(g["H"])
        kv = bf16_from_u16(g["kv_buffer_after"]).to(dev()).contiguous()          # [slots,1,576]
        pages = kv.shape[0] // 64
        q = bf16_from_u16(g["q"]).to(dev()).view(-1, 1, H, 576).contiguous()
        seq = torch.from_numpy(g["seq_lens"]).to(dev())
        bt = torch.from_numpy(g["block_table"]).to(dev())
        meta, ns = flash_mla_swap.get_mla_metadata(seq, H, 1)
        o, _ = flash_mla_swap.flash_mla_with_kvcache(q, kv.view(pages, 64, 1, 576), bt, seq, 512, meta, ns,
                                                     float(g["scaling"]), True)
        ref = bf16_from_u16(g["o"]).view(-1, 1, H, 512)
        r = rel_mae(o.cpu(), ref)
        assert r < 8e-3, (name, r)   # both sides round P and the output to bf16 (different summation orders)


def test_quant_divisions_bit_exact_on_adversarial_rows(fm):
    """The quantise kernels divide by a shared scale through fl_div8_to_fp8 (csrc/fl_common.h) instead of the compiler's
    per-element IEEE division: 2.4 M elements with row magnitudes over 36 decades, exact and signed zeros, values tiny
    next to their row's maximum, rows of zeros — K5 / K4 bytes, scales and rope bits identical to the torch statement."""
    g = torch.Generator().manual_seed(77)
    n, slots = 4096, 4096
    key = torch.randn(n, 1, 576, generator=g) * torch.pow(10.0, torch.rand(n, 1, 1, generator=g) * 36 - 18)
    key = key * torch.pow(2.0, -torch.randint(0, 40, (n, 1, 576), generator=g).float())      # wide range inside a row
    key[torch.rand(n, 1, 576, generator=g) < 0.05] = 0.0
    key[torch.rand(n, 1, 576, generator=g) < 0.05] = -0.0
    key[::97] = 0.0
    key = key.to(torch.bfloat16)
    loc = torch.randperm(slots, generator=g)[:n].to(torch.int32)
    ref = [torch.zeros(slots, 1, 512, dtype=torch.uint8), torch.zeros(slots, 1, 1), torch.zeros(slots, 1, 64, dtype=torch.bfloat16)]
    mla_ref.quantize_and_cache_k(key, ref[0], ref[1], ref[2], loc)
    out = [torch.zeros_like(t, device=dev()) for t in ref]
    fm.quantize_and_cache_k(key.to(dev()), out[0], out[1], out[2], loc.to(dev()), 512)
    assert torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[1].cpu(), ref[1])
    assert torch.equal(out[2].cpu().view(torch.int16), ref[2].view(torch.int16))
    q = key.view(32, 1, 128, 576).contiguous()
    qn, qs, qr = fm.quantize_ckv_per_token_head(q.to(dev()), 512)
    rn, rs, rr = mla_ref.quantize_ckv_per_token_head(q, 512)
    assert torch.equal(qn.cpu().view(torch.uint8), rn.view(torch.uint8)) and torch.equal(qs.cpu(), rs)
    assert torch.equal(qr.cpu().view(torch.int16), rr.view(torch.int16))


def test_zz_write_measured_errors():
    """Not a check: dumps the per-case measured errors of this run (rel-MAE, max-abs, LSE) so that the stated tolerances
    can be read against what the kernel actually does (profiles/r02_mla_parity_measured.json is a committed copy)."""
    import json
    import os

    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "mla_parity_measured.json"), "w") as f:
            json.dump(MEASURED, f, indent=1)
    except OSError:
        pass
    if MEASURED:
        worst = max(MEASURED, key=lambda m: m["rel_mae"])
        print(f"\nK1 measured over {len(MEASURED)} cases: worst rel-MAE {worst['rel_mae']:.4f} ({worst['case']})")


def test_replay_reference_call_trace(fm):
    """Replays every call FlashMLABackend makes into flash_mla_fp8 / flash_mla_swap (recorded from the reference source:
    tests/golden/flashmla_backend_call_trace.json) through the real drop-in modules with tensors of exactly the recorded
    shapes, dtypes and strides; the result must support the view the backend takes of it."""
    import json
    import os

    import flash_mla_swap

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flashmla_backend_call_trace.json")) as f:
        trace = json.load(f)["trace"]
    mods = {"flash_mla_fp8": fm, "flash_mla_swap": flash_mla_swap}

    def make(d, fill=None):
        if not d.get("tensor"):
            return d.get("value")
        dt = getattr(torch, d["dtype"])
        n = 1 + sum((s - 1) * st for s, st in zip(d["shape"], d["stride"])) if all(s > 0 for s in d["shape"]) else 0
        base = torch.zeros(max(n, 1), dtype=torch.float32, device=dev())
        if fill is not None:
            base.fill_(fill)
        elif dt in (torch.bfloat16, torch.float32):
            base.normal_()
        return torch.as_strided(base.to(dt), d["shape"], d["stride"])

    for case in trace:
        H, bs, s_q = case["H"], case["bs"], case["s_q"]
        quantised = None
        for call in case["calls"]:
            if call["module"] not in mods:
                continue
            fn = getattr(mods[call["module"]], call["fn"])
            if call["fn"] == "quantize_ckv_per_token_head":
                q = make(call["args"][0])
                quantised = fn(q, call["args"][1]["value"])
                assert [tuple(t.shape) for t in quantised] == [(bs, s_q, H, 512), (bs, s_q, H, 1), (bs, s_q, H, 64)]
                continue
            kw = {}
            for k, d in call["kwargs"].items():
                if k == "cache_seqlens":
                    kw[k] = make(d, fill=100)
                elif k in ("block_table", "tile_scheduler_metadata", "num_splits"):
                    kw[k] = make(d, fill=0)
                elif k in ("descale_q", "descale_k"):
                    kw[k] = make(d, fill=1)
                elif k == "k_scale":
                    kw[k] = make(d, fill=1)
                else:
                    kw[k] = make(d)
            if quantised is not None:
                kw["q_nope"], kw["q_scale"], kw["q_rope"] = quantised
                quantised = None
            # the backend passes persistent metadata buffers it filled from get_mla_metadata (flashmla_backend.py:307-321)
            m = mods[call["module"]].get_mla_metadata(kw["cache_seqlens"], s_q * H, 1)
            kw["tile_scheduler_metadata"], kw["num_splits"] = m
            o, lse = fn(**kw)
            torch.cuda.synchronize()
            assert o.shape == (bs, s_q, H, 512) and o.dtype == torch.bfloat16 and lse.shape == (bs, H, s_q), case["case"]
            assert torch.isfinite(o.float()).all(), case["case"]
            assert list(o.view(-1, H * 512).shape) == case["returned"]["shape"], case["case"]


def test_split_merge_inside_the_decode_kernel_on_every_split_case():
    """The role-specialised mapping merges split requests INSIDE the decode kernel (the request's first piece waits for the
    others' arrival counts and merges from its registers, mla_decode_fp8_y.hip) when there are at least half as many requests
    as parts; smaller batches take the merge kernel.  Here the parity cases, the
    reference-jump cases and the graph-replay test of this file run once more in a process that forces the in-kernel merge
    for every shape (FLUENT_MLA_MERGE_KERNEL=0 is read once per process), so that its counters (spare metadata columns,
    reset by the merging part), the many-way splits (ns up to 128) and replays on unchanged metadata are all exercised."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, FLUENT_MLA_MERGE_KERNEL="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "decode_parity_vs_oracle or reference_jump or mtp_verify_and_draft or full_size_properties_cfg2_ragged"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1500:]


@pytest.mark.parametrize("beside_gemm", ["0", "1"])
def test_in_kernel_split_merge_is_deterministic_and_makes_progress_beside_a_gemm(beside_gemm):
    """tools/determinism_ragged.py as a test: the ragged cfg2 workload (~122 requests split in two, merged inside the
    decode kernel by their first piece, which polls the other pieces' arrival counter) launched 120 times on ONE metadata tensor — every
    output bit-identical to the first, the counters back to zero.  BESIDE_GEMM=1: a compute-regime grouped GEMM (one 8-wave workgroup per
    CU) owns the CUs on a second stream the whole time, so the decode workgroups are dispatched late and out of step and the merging piece
    really waits for pieces that have not started: forward progress of the bounded poll under CU contention (run under a timeout)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "determinism_ragged.py"), "120"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, BESIDE_GEMM=beside_gemm))
    assert r.returncode == 0, r.stderr[-1500:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["beside_gemm"] == (beside_gemm == "1") and rec["iterations"] == 120 and rec["split_requests"] > 100
    assert rec["mismatching_launches"] == 0 and rec["merge_counters_back_to_zero"], rec


@pytest.mark.gpu
def test_in_kernel_split_merge_timeout_is_reported_not_silent(fm):
    """a merger that gives up waiting for a request's other pieces writes NaN into the request's rows AND the next decode call
    raises.  Forced here by handing the kernel scheduler metadata whose second part does no work (its pieces never arrive)."""
    from fluent_mi355 import mla as hip_mla
    H, bs = 128, 64                                   # 64 requests on 128 parts: every request is cut in two and merged inside the kernel
    c = make_paged_case([4096] * bs, H, 1, seed=3)
    d = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in c.items()}
    pages = c["total_pages"]
    qn, qs, qr = fm.quantize_ckv_per_token_head(d["q"].contiguous(), 512)

    def decode(meta, ns):
        return fm.flash_mla_ckv_fp8_per_token(
            q_nope=qn, q_rope=qr, k_cache_lora=d["k_lora"].view(pages, 64, 1, 512), k_cache_rope=d["k_rope"].view(pages, 64, 1, 64),
            q_scale=qs, k_scale=d["k_scale"].view(pages, 64, 1, 1), block_table=d["block_table"], cache_seqlens=d["cache_seqlens"],
            head_dim_v=512, tile_scheduler_metadata=meta, num_splits=ns, softmax_scale=SCALE, causal=True)

    meta, ns = fm.get_mla_metadata(d["cache_seqlens"], H, 1)
    good, _ = decode(meta, ns)
    torch.cuda.synchronize()
    assert int(ns[1]) == 2 and int(meta[1, 0]) == 0      # request 0 = parts 0 and 1
    bad = meta.clone()
    bad[1, 0:4] = torch.tensor([bs, 0, bs, 0], dtype=torch.int32)   # part 1 does nothing: request 0's second piece never arrives
    hip_mla.set_merge_timeout(0.05)
    try:
        o, lse = decode(bad, ns)
        torch.cuda.synchronize()
        assert torch.isnan(o[0].float()).all() and torch.isnan(lse[0]).all()          # poisoned, not plausible numbers
        assert torch.equal(o[2:].view(torch.int16), good[2:].view(torch.int16))        # the other requests are untouched
        with pytest.raises(RuntimeError, match="gave up waiting"):
            decode(meta, ns)
    finally:
        hip_mla.set_merge_timeout(2.0)
    meta2, ns2 = fm.get_mla_metadata(d["cache_seqlens"], H, 1)                        # rebuilt metadata: business as usual
    again, _ = decode(meta2, ns2)
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int16), good.view(torch.int16))
