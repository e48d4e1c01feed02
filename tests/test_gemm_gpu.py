"""GPU parity tests of the FP8 block-scaled grouped GEMM / quant / activation path (pytest -m gpu), through the
drop-in modules (`deep_gemm`, `flashinfer`, `eps`) -> ctypes -> C-ABI -> HIP kernels, against the oracle
(oracle/gemm_ref.py = python/sglang/test/test_block_fp8.py restated) and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from helpers import bf16_from_u16, fp8_from_u8, load_golden, rel_mae
from oracle import gemm_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def fp8_weights(g, *shape):
    return ((torch.rand(*shape, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(torch.float8_e4m3fn)


# ------------------------------------------------------------------ Q1/Q2 + A1: bit-exact byte work
def test_quant_1x128_bit_exact_vs_reference_golden():
    import flashinfer

    g = load_golden("gemm_block_fp8.npz")
    x = bf16_from_u16(g["quant_x"]).to(DEV)
    M, K = x.shape
    xq = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV)
    xs = torch.empty(M, K // 128, dtype=torch.float32, device=DEV)
    flashinfer.sgl_per_token_group_quant_fp8(x, xq, xs, 128, 1e-10, -448.0, 448.0, False)
    assert np.array_equal(xq.cpu().view(torch.uint8).numpy(), g["quant_q"])
    assert np.array_equal(xs.cpu().numpy().view(np.uint32), g["quant_s"].view(np.uint32))
    # the MoE executor's column-major padded scale layout (fp8_eps_executor.py:52-55)
    E, mp = 4, (M + 4 * 31) // 32 * 32
    xs2 = torch.zeros((K // 128, mp), dtype=torch.float32, device=DEV).permute(-1, -2)
    xq2 = torch.empty_like(xq)
    flashinfer.quantization.quant_1x128(x, xq2, xs2, torch.zeros(E + 1, dtype=torch.int32, device=DEV), E, (M + 3) // 4 * 4, mp, K)
    assert torch.equal(xq2.view(torch.uint8), xq.view(torch.uint8)) and torch.equal(xs2[:M].contiguous(), xs)


def test_quant_1x128_divisions_bit_exact_on_adversarial_groups():
    """Q1 divides by the group scale through fl_div8_to_fp8 (csrc/fl_common.h): 4 M elements with group magnitudes over 30
    decades, signed and exact zeros, values tiny next to their group's maximum, all-zero groups (eps clamp) — bytes and scales
    identical to the reference statement (test_block_fp8.py:15-40 restated in oracle.gemm_ref)."""
    import flashinfer

    g = torch.Generator().manual_seed(78)
    M, K = 2048, 2048
    x = torch.randn(M, K // 128, 128, generator=g) * torch.pow(10.0, torch.rand(M, K // 128, 1, generator=g) * 30 - 15)
    x = x * torch.pow(2.0, -torch.randint(0, 40, (M, K // 128, 128), generator=g).float())
    x[torch.rand(M, K // 128, 128, generator=g) < 0.05] = 0.0
    x[torch.rand(M, K // 128, 128, generator=g) < 0.05] = -0.0
    x[::53] = 0.0
    x = x.view(M, K).to(torch.bfloat16)
    rq, rs = gemm_ref.per_token_group_quant_fp8(x, 128)
    xq = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV)
    xs = torch.empty(M, K // 128, dtype=torch.float32, device=DEV)
    flashinfer.sgl_per_token_group_quant_fp8(x.to(DEV), xq, xs, 128, 1e-10, -448.0, 448.0, False)
    assert torch.equal(xq.cpu().view(torch.uint8), rq.view(torch.uint8))
    assert torch.equal(xs.cpu().view(torch.int32), rs.view(torch.int32))


def test_silu_and_mul_bit_exact_vs_reference_golden():
    import flashinfer
    from eps.executor import silu

    g = load_golden("gemm_block_fp8.npz")
    x = bf16_from_u16(g["silu_x"]).to(DEV)
    ref = bf16_from_u16(g["silu_out"])
    out = torch.empty(x.shape[0], x.shape[1] // 2, dtype=torch.bfloat16, device=DEV)
    flashinfer.silu_and_mul(x, out)
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
    assert torch.equal(silu(x, None, 0).cpu().view(torch.int16), ref.view(torch.int16))
    # fused silu + 1x128 quant == oracle quant of the bf16 result
    gen = torch.Generator().manual_seed(3)
    y = (torch.randn(37, 512, generator=gen) * 2).to(torch.bfloat16)
    q = torch.empty(37, 256, dtype=torch.float8_e4m3fn, device=DEV)
    s = torch.empty(37, 2, dtype=torch.float32, device=DEV)
    flashinfer.activation.silu_and_mul_fuse_block_quant(y.to(DEV), s, q, True)
    rq, rs = gemm_ref.per_token_group_quant_fp8(gemm_ref.silu_and_mul(y), 128)
    assert torch.equal(q.cpu().view(torch.uint8), rq.view(torch.uint8)) and torch.equal(s.cpu(), rs)


# ------------------------------------------------------------------ G4 dense
def test_dense_gemm_vs_reference_golden():
    import deep_gemm

    g = load_golden("gemm_block_fp8.npz")
    A, B = fp8_from_u8(g["mm_A"]).to(DEV), fp8_from_u8(g["mm_B"]).to(DEV)
    As, Bs = torch.from_numpy(g["mm_As"]).to(DEV), torch.from_numpy(g["mm_Bs"]).to(DEV)
    C = torch.empty(A.shape[0], B.shape[0], dtype=torch.bfloat16, device=DEV)
    deep_gemm.gemm_fp8_fp8_bf16_nt((A, As), (B, Bs), C, True)
    # reference threshold: rel-MAE < 1e-3 (python/sglang/test/test_block_fp8.py:185-189)
    assert rel_mae(C.cpu(), bf16_from_u16(g["mm_C"])) < 1e-3


@pytest.mark.parametrize("M,N,K", [(1, 128, 128), (5, 2112, 7168), (130, 384, 512), (257, 7168, 2048), (64, 260, 256)])
def test_dense_gemm_shapes(M, N, K):
    import deep_gemm
    from fluent_mi355.gemm import per_token_group_quant_fp8

    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = fp8_weights(g, N, K)
    Ws = torch.rand((N + 127) // 128, K // 128, generator=g) * 1e-2
    xq, xs = per_token_group_quant_fp8(x.to(DEV), column_major_scales=True)   # TMA-aligned col-major like deep_geem.py
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    deep_gemm.gemm_fp8_fp8_bf16_nt((xq, xs), (W.to(DEV), Ws.to(DEV)), out)
    rq, rs = gemm_ref.per_token_group_quant_fp8(x, 128)
    assert torch.equal(xq.cpu().view(torch.uint8), rq.view(torch.uint8)) and torch.equal(xs.cpu().contiguous(), rs)
    ref = gemm_ref.block_fp8_matmul(rq, W, rs, Ws)
    assert rel_mae(out.cpu(), ref) < 1e-3, (M, N, K)


# ------------------------------------------------------------------ G1 offset / G2 contiguous / G3 masked
def make_group_case(counts, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    E, M = len(counts), sum(counts)
    x = (torch.randn(M, K, generator=g) / 3).to(torch.bfloat16)
    xq, xs = gemm_ref.per_token_group_quant_fp8(x, 128)
    W = fp8_weights(g, E, N, K)
    Ws = torch.rand(E, (N + 127) // 128, K // 128, generator=g) * 1e-2
    ex = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32)
    return xq, xs, W, Ws, ex


@pytest.mark.parametrize("counts,N,K", [([4, 0, 7, 1], 256, 512), ([0, 0, 33, 0, 64, 1, 200], 512, 256),
                                        ([3] * 32, 4096, 7168), ([150, 70], 7168, 2048), ([0, 0, 0], 128, 128),
                                        # >= 128 rows per expert: several 128-token tiles per expert
                                        ([300, 0, 129, 511], 768, 512), ([512, 512], 4096, 7168), ([129, 640, 1], 260, 256)])
def test_grouped_offset_vs_oracle(counts, N, K):
    import deep_gemm

    xq, xs, W, Ws, ex = make_group_case(counts, N, K, seed=len(counts) + N)
    M, E = xq.shape[0], len(counts)
    mp = (M + E * 31) // 32 * 32 + 32
    xs_dev = torch.zeros((K // 128, mp), dtype=torch.float32, device=DEV).permute(-1, -2)   # executor's layout
    xs_dev[:M] = xs.to(DEV)
    out = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device=DEV)     # rows past ex[-1] must stay untouched
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq.to(DEV), xs_dev[:M]), (W.to(DEV), Ws.to(DEV)), out[:M], ex.to(DEV),
                                                    use_pdl=True)
    ref = gemm_ref.grouped_gemm_offset(xq, xs, W, Ws, ex)
    if M:
        assert rel_mae(out[:M].cpu(), ref) < 1e-3
    assert bool((out[M:] == 7.0).all())


def _route_counts(T, E, top_k, zipf, seed):
    """rows per expert of a top-k routing without replacement per token: uniform, or Zipf-skewed expert popularity"""
    g = torch.Generator().manual_seed(seed)
    p = torch.ones(E) if not zipf else 1.0 / torch.arange(1, E + 1, dtype=torch.float64) ** zipf
    p = p[torch.randperm(E, generator=g)]
    ids = torch.multinomial((p / p.sum()).expand(T, E), top_k, replacement=False, generator=g)
    return torch.bincount(ids.reshape(-1), minlength=E).tolist()


@pytest.mark.parametrize("T,zipf,N,K", [(32, 0.0, 256, 256),       # decode: 256 rows over 256 experts, many empty
                                        (160, 0.0, 256, 256),      # ~5 rows per expert
                                        (160, 1.2, 256, 512),      # skewed: a few hot experts, most empty
                                        (2048, 1.2, 384, 256),     # skewed, hot experts with > 256 rows (several tiles each)
                                        (4096, 0.0, 256, 256)])    # 128 rows per expert (BASELINE config 3 at T = 4096)
def test_grouped_offset_e256_top8_routing_vs_oracle(T, zipf, N, K):
    """BASELINE config 3's group structure (E = 256 experts, top-8 routing incl. empty experts and a Zipf-skewed load): the
    offset-mode tile lookup walks the groups 64 at a time with a running base (grouped_gemm_shared.h locate_tile), and only
    E > 64 exercises iterations 2..4 of that walk."""
    import deep_gemm

    counts = _route_counts(T, 256, 8, zipf, seed=T + int(zipf * 10))
    assert sum(counts) == T * 8 and len(counts) == 256
    if T <= 160:
        assert counts.count(0) > 0   # empty experts are part of the case
    xq, xs, W, Ws, ex = make_group_case(counts, N, K, seed=T)
    M = xq.shape[0]
    mp = (M + 256 * 31) // 32 * 32 + 32
    xs_dev = torch.zeros((K // 128, mp), dtype=torch.float32, device=DEV).permute(-1, -2)
    xs_dev[:M] = xs.to(DEV)
    out = torch.full((M + 5, N), 7.0, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq.to(DEV), xs_dev[:M]), (W.to(DEV), Ws.to(DEV)), out[:M], ex.to(DEV))
    ref = gemm_ref.grouped_gemm_offset(xq, xs, W, Ws, ex)
    o = out[:M].cpu()
    assert rel_mae(o, ref) < 1e-3
    # per-expert check: a wrong running base would put a tile's rows under the wrong expert's weights
    exl = ex.tolist()
    for e in (0, 63, 64, 65, 127, 128, 191, 192, 255):
        lo, hi = exl[e], exl[e + 1]
        if hi > lo:
            assert rel_mae(o[lo:hi], ref[lo:hi]) < 2e-3, e
    assert bool((out[M:] == 7.0).all())


@pytest.mark.parametrize("T,zipf", [(128, 0.0), (4096, 0.0), (4096, 1.2)])
def test_cfg3_full_size_sampled_experts_vs_oracle(T, zipf):
    """BASELINE config 3 at FULL size — E = 256 experts x w13 [4096, 7168] (7.5 GB of fp8 weights) and w2 [7168, 2048], top-8
    routing of T tokens (decode: 4 rows per expert; 4096: 128 rows per expert -> the 128- and, with the Zipf load, the
    256-row tiles) — the shape bench.py times, which the small-N/K E=256 tests and the E<=32 full-N/K tests only bracket.
    Weights and activations are generated on the device; six sampled experts (first, last, the 64-group boundaries of the tile
    lookup, the most and the least loaded non-empty one) are copied back and recomputed by oracle.gemm_ref."""
    import deep_gemm

    E, K1, N1, N2 = 256, 7168, 4096, 7168
    counts = _route_counts(T, E, 8, zipf, seed=31 + T + int(zipf * 10))
    ex = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32)
    M = int(ex[-1])
    g = torch.Generator(device=DEV).manual_seed(T + 1)

    def dev_fp8(*shape):   # uniform random e4m3 bytes without the NaN patterns
        b = torch.randint(0, 255, shape, device=DEV, generator=g, dtype=torch.int16)
        return torch.where((b & 0x7F) == 0x7F, b - 1, b).to(torch.uint8).view(torch.float8_e4m3fn)

    nonempty = [e for e in range(E) if counts[e] > 0]
    picks = sorted({nonempty[0], nonempty[-1], min(nonempty, key=lambda e: abs(e - 63.5)), min(nonempty, key=lambda e: abs(e - 128)),
                    max(nonempty, key=lambda e: counts[e]), min(nonempty, key=lambda e: counts[e])})
    mp = (M + E * 31) // 32 * 32 + 32
    for (N, K) in ((N1, K1), (N2, N1 // 2)):          # w13: [4096, 7168]; w2: [7168, 2048]
        x = (torch.randn(M, K, device=DEV, generator=g) / 3).to(torch.bfloat16)
        xq_ref, xs_ref = gemm_ref.per_token_group_quant_fp8(x.cpu(), 128)
        W = dev_fp8(E, N, K)
        Ws = torch.rand(E, N // 128, K // 128, device=DEV, generator=g) * 1e-2
        xs_dev = torch.zeros((K // 128, mp), dtype=torch.float32, device=DEV).permute(-1, -2)
        xs_dev[:M] = xs_ref.to(DEV)
        out = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device=DEV)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq_ref.to(DEV), xs_dev[:M]), (W, Ws), out[:M], ex.to(DEV))
        torch.cuda.synchronize()
        assert bool((out[M:] == 7.0).all()) and bool(torch.isfinite(out[:M].float()).all())
        for e in picks:
            lo, hi = int(ex[e]), int(ex[e + 1])
            ref = gemm_ref.block_fp8_matmul(xq_ref[lo:hi], W[e].cpu(), xs_ref[lo:hi], Ws[e].cpu())
            assert rel_mae(out[lo:hi].cpu(), ref) < 1e-3, (N, K, e, hi - lo)
        del W, Ws, out, x


def test_grouped_contiguous_and_masked_vs_oracle():
    import deep_gemm

    counts = [128, 256, 0, 128]
    N, K = 384, 512
    xq, xs, W, Ws, ex = make_group_case(counts, N, K, seed=9)
    M = xq.shape[0]
    m_indices = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts)).to(torch.int32)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((xq.to(DEV), xs.to(DEV)), (W.to(DEV), Ws.to(DEV)), out, m_indices.to(DEV), True)
    ref = gemm_ref.grouped_gemm_contiguous(xq, xs, W, Ws, m_indices)
    assert rel_mae(out.cpu(), ref) < 1e-3
    # masked: [G, Mp, K] with only the first masked_m[g] rows valid; NaN-poison the padding rows of A
    G, Mp = 3, 96
    masked = torch.tensor([5, 0, 96], dtype=torch.int32)
    g = torch.Generator().manual_seed(4)
    xm = (torch.randn(G, Mp, K, generator=g) / 3).to(torch.bfloat16)
    aq, asc = gemm_ref.per_token_group_quant_fp8(xm, 128)
    aq_dev = aq.clone().view(torch.uint8)
    for gi in range(G):
        aq_dev[gi, int(masked[gi]):] = 0x7F
    om = torch.full((G, Mp, N), 3.0, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_masked((aq_dev.to(DEV).view(torch.float8_e4m3fn), asc.to(DEV)), (W[:G].to(DEV), Ws[:G].to(DEV)),
                                                    om, masked.to(DEV), 32, True)
    refm = gemm_ref.grouped_gemm_masked(aq, asc, W[:G], Ws[:G], masked)
    for gi in range(G):
        mm = int(masked[gi])
        if mm:
            assert rel_mae(om[gi, :mm].cpu(), refm[gi, :mm]) < 1e-3
        assert bool((om[gi, mm:] == 3.0).all())


def test_moe_pipeline_vs_reference_golden():
    """Q1 -> G1(w13) -> A1 -> Q1 -> G1(w2) -> weighted sum (fp8_eps_executor.py:33-82) vs the reference's
    torch_w8a8_block_fp8_moe output (python/sglang/test/test_block_fp8.py:212-241); threshold 2e-2 (:310-314)."""
    import deep_gemm
    import flashinfer
    from eps.executor import silu

    g = load_golden("gemm_block_fp8.npz")
    a = bf16_from_u16(g["moe_a"])
    w1, w2 = fp8_from_u8(g["moe_w1"]), fp8_from_u8(g["moe_w2"])
    w1s, w2s = torch.from_numpy(g["moe_w1_s"]), torch.from_numpy(g["moe_w2_s"])
    tw, ti = torch.from_numpy(g["moe_topk_w"]), torch.from_numpy(g["moe_topk_ids"]).long()
    B, D = a.shape
    E, topk = w1.shape[0], ti.shape[1]
    flat = ti.reshape(-1)
    order = torch.argsort(flat, stable=True)
    rows = a.repeat_interleave(topk, dim=0)[order].contiguous().to(DEV)       # routed rows sorted by expert
    ex = torch.zeros(E + 1, dtype=torch.int32)
    ex[1:] = torch.cumsum(torch.bincount(flat, minlength=E), 0)
    M = rows.shape[0]
    mp = (M + E * 31) // 32 * 32

    def q(x, K):
        xq = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV)
        xs = torch.empty((K // 128, mp), dtype=torch.float32, device=DEV).permute(-1, -2)
        flashinfer.quantization.quant_1x128(x, xq, xs, ex.to(DEV), E, (M + 3) // 4 * 4, mp, K)
        return xq, xs

    gate_up = torch.empty(M, w1.shape[1], dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset(q(rows, D), (w1.to(DEV), w1s.to(DEV)), gate_up, ex.to(DEV), use_pdl=True)
    act = silu(gate_up, ex.to(DEV), M)
    down = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset(q(act, w2.shape[2]), (w2.to(DEV), w2s.to(DEV)), down, ex.to(DEV), use_pdl=True)
    unsorted = torch.empty_like(down)
    unsorted[order.to(DEV)] = down
    out = (unsorted.view(B, topk, D) * tw.to(DEV).view(B, topk, 1).to(torch.bfloat16)).sum(dim=1)
    assert rel_mae(out.cpu(), bf16_from_u16(g["moe_out"])) < 2e-2


def test_gemm_graph_capture():
    import deep_gemm

    xq, xs, W, Ws, ex = make_group_case([9, 0, 40], 256, 256, seed=2)
    args = ((xq.to(DEV), xs.to(DEV)), (W.to(DEV), Ws.to(DEV)))
    out = torch.zeros(xq.shape[0], 256, dtype=torch.bfloat16, device=DEV)
    exd = ex.to(DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset(*args, out, exd)
    eager = out.clone()
    out.zero_()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset(*args, out, exd)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset(*args, out, exd)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_silu_and_mul_masked_post_quant_vs_oracle():
    """A1-masked (deep_ep_executor.py:106-170): [G, M, 2I] with masked_m rows per group in ONE launch; bytes and scales of the
    live rows identical to the A1 + Q1 statements (oracle.gemm_ref), rows beyond masked_m never written (poisoned before);
    also through flashinfer.activation.silu_and_mul_fuse_block_quant(masked_m=...) and with column-major scales."""
    import flashinfer
    from fluent_mi355.gemm import silu_and_mul_masked_post_quant_fwd

    G, M, I = 5, 96, 384
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(G, M, 2 * I, generator=g) * 2).to(torch.bfloat16)
    masked = torch.tensor([96, 0, 17, 1, 64], dtype=torch.int32)
    act = gemm_ref.silu_and_mul(x.view(G * M, 2 * I)).view(G, M, I)
    rq, rs = gemm_ref.per_token_group_quant_fp8(act.contiguous(), 128)
    for col_major in (False, True):
        q = torch.full((G, M, I), 0x7F, dtype=torch.uint8, device=DEV).view(torch.float8_e4m3fn)
        sc = torch.full((G, I // 128, M) if col_major else (G, M, I // 128), -7.0, device=DEV)
        sc = sc.permute(0, 2, 1) if col_major else sc
        if col_major:
            flashinfer.activation.silu_and_mul_fuse_block_quant(x.to(DEV), sc, q, True, masked_m=masked.to(DEV))
        else:
            silu_and_mul_masked_post_quant_fwd(x.to(DEV), q, sc, 128, masked.to(DEV))
        qc, scc = q.view(torch.uint8).cpu(), sc.cpu()
        for gi in range(G):
            m = int(masked[gi])
            assert torch.equal(qc[gi, :m], rq.view(torch.uint8)[gi, :m]) and torch.equal(scc[gi, :m], rs[gi, :m])
            assert bool((qc[gi, m:] == 0x7F).all()) and bool((scc[gi, m:] == -7.0).all())


def test_ep_all_to_all_single_rank_hip_row_ops():
    """C1/C2 device kernels (route / sort / gather / scatter / combine) on one GPU (world 1: the exchange is a copy);
    the multi-rank host logic is covered on CPU with gloo in test_ep_gloo_cpu.py."""
    from eps.fast_ep import AllToAll

    E, K, HID, t = 16, 4, 256, 37
    g = torch.Generator().manual_seed(1)
    x = torch.randn(t, HID, generator=g).to(torch.bfloat16)
    idx = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(t)]).to(torch.int32)
    w = torch.rand(t, K, generator=g)
    a2a = AllToAll(K, E, HID, 64, None)
    ex = torch.empty(E + 1, dtype=torch.int32, device=DEV)
    expert_x = torch.zeros(t * K, HID, dtype=torch.bfloat16, device=DEV)
    a2a.dispatch(out_exclusive_sum=ex, out_expert_x=expert_x, dp_x=x.to(DEV), indices=idx.to(DEV), num_global_tokens=t)
    counts = torch.bincount(idx.reshape(-1).long(), minlength=E)
    assert ex.cpu().tolist() == [0] + torch.cumsum(counts, 0).tolist()
    exc = ex.cpu()
    y = torch.full_like(expert_x, float("nan"))
    for e in range(E):
        y[int(exc[e]):int(exc[e + 1])] = (expert_x[int(exc[e]):int(exc[e + 1])].float() * (e + 1)).to(torch.bfloat16)
        # every row grouped under expert e is a token that selected e
    out = torch.empty(t, HID, dtype=torch.bfloat16, device=DEV)
    a2a.combine(out_tokens=out, weights=w.to(DEV), expert_y=y, num_global_tokens=t)
    ref = sum(w[:, k:k + 1] * (x.float() * (idx[:, k:k + 1].float() + 1)).to(torch.bfloat16).float() for k in range(K)).to(torch.bfloat16)
    assert torch.allclose(out.cpu().float(), ref.float(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("layout", ["dense", "message_tail"])
def test_ep_dedup_row_kernels_match_the_torch_row_ops_world4_routing(layout):
    """The token-once-per-peer routing kernels (route_dedup / gather_f32 / sort with inverse / gather_div / send) against the
    torch-indexing implementation the gloo tests inject, with the routing of a 4-rank job (no exchange needed to compare the
    kernels) — on dense [rows, top_k] tensors and on VIEWS into a slab-row message [rows, hidden | ids | weights] (the
    single-message dispatch: the kernels take row strides)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ep_torch_ops import TorchRowOps
    from fluent_mi355.ep import HipRowOps

    hip, ref = HipRowOps(), TorchRowOps()
    W, EPR, K, T, HID = 4, 8, 8, 53, 128
    cap = 40                                                   # < T on purpose: rows past the slab end are dropped the same way
    S = W * cap
    g = torch.Generator().manual_seed(77)
    idx = torch.stack([torch.randperm(W * EPR, generator=g)[:K] for _ in range(T)]).to(torch.int32)
    idx[3, 2] = -1                                              # an invalid id is ignored by both
    x = torch.randn(T, HID, generator=torch.Generator().manual_seed(8)).to(torch.bfloat16)
    outs = []
    for ops, dev in ((ref, "cpu"), (hip, DEV)):
        ts = torch.empty(T * W, dtype=torch.int32, device=dev)
        pp = torch.empty(S * K, dtype=torch.int32, device=dev)   # pair_src: the (token, k) behind every slab pair
        if layout == "dense":
            rows = torch.zeros(S, HID, dtype=torch.bfloat16, device=dev)
            se = torch.empty(S, K, dtype=torch.int32, device=dev)
            placed = torch.empty(S, K, dtype=torch.float32, device=dev)
        else:
            msg = torch.zeros(S, HID + 4 * K, dtype=torch.bfloat16, device=dev)
            rows, se = msg[:, :HID], msg.view(torch.int32)[:, HID // 2:HID // 2 + K]
            placed = msg.view(torch.float32)[:, HID // 2 + K:HID // 2 + 2 * K]
        ops.route_dedup(idx.to(dev).reshape(-1), K, EPR, W, cap, ts, se, pp)
        ops.send(x.to(dev), ts, W, rows)
        vals = torch.rand(T * K, generator=torch.Generator().manual_seed(5)).to(dev)
        ops.gather_f32(vals, pp, placed)
        # the same weights placed by the routing launch itself (dispatch(..., weights=): no gather launch)
        placed2 = torch.full((S, K), 7.0, dtype=torch.float32, device=dev)
        ops.route_dedup(idx.to(dev).reshape(-1), K, EPR, W, cap, torch.empty_like(ts), torch.empty(S, K, dtype=torch.int32, device=dev),
                        torch.empty_like(pp), vals, placed2)
        assert torch.equal(placed2.cpu(), placed.cpu().contiguous())
        order = torch.empty(S * K, dtype=torch.int32, device=dev)
        ex = torch.empty(EPR + 1, dtype=torch.int32, device=dev)
        inv = torch.full((S * K,), -1, dtype=torch.int32, device=dev)
        ops.sort(se, EPR, order, ex, inv)
        n = int(ex[-1])
        dst = torch.zeros(S * K, HID, dtype=torch.bfloat16, device=dev)
        ops.gather_div(rows, order, S * K, K, dst, ex[EPR:])           # static launch over the row bound, device-side row count
        assert bool((dst[n:] == 0).all())                               # rows past exclusive_sum[-1] are not touched
        dst = dst[:n]
        outs.append([t.cpu().contiguous() for t in (ts, pp, se, placed, ex, rows)])
        # the order INSIDE an expert group is free (the device sort hands positions out with atomics): check it by meaning
        o, sec, exc = order.cpu().long(), se.cpu().reshape(-1), ex.cpu()
        assert sorted(o.tolist()) == list(range(S * K))
        for e in range(EPR):
            assert bool((sec[o[int(exc[e]):int(exc[e + 1])]] == e).all())
        assert bool((sec[o[n:]] < 0).all())
        assert torch.equal(inv.cpu().long()[o], torch.arange(S * K))
        assert torch.equal(dst.cpu(), rows.cpu()[o[:n] // K])
        # weighted return with the weights read through the (possibly strided) view
        back = torch.empty(S, HID, dtype=torch.bfloat16, device=dev)
        y = torch.full((S * K, HID), 2.0, dtype=torch.bfloat16, device=dev)   # (position-free values: the order inside a group differs between the two)
        ops.combine(y, inv, placed, back, K)
        outs[-1].append(back.float().cpu())
    for a, b in zip(outs[0][:-1], outs[1][:-1]):               # slab positions are handed out in token order by both
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
    assert torch.allclose(outs[0][-1], outs[1][-1], rtol=1e-2, atol=1e-2)   # expert-side weighted sums through the weight view


def test_masked_big_tile_vs_oracle():
    """Masked mode with >= 128 expected rows per group (128-token tiles); padding rows NaN-poisoned and left untouched."""
    import deep_gemm

    G, Mp, N, K = 3, 384, 512, 512
    masked = torch.tensor([384, 0, 130], dtype=torch.int32)
    g = torch.Generator().manual_seed(14)
    xm = (torch.randn(G, Mp, K, generator=g) / 3).to(torch.bfloat16)
    aq, asc = gemm_ref.per_token_group_quant_fp8(xm, 128)
    W = fp8_weights(g, G, N, K)
    Ws = torch.rand(G, N // 128, K // 128, generator=g) * 1e-2
    aq_dev = aq.clone().view(torch.uint8)
    for gi in range(G):
        aq_dev[gi, int(masked[gi]):] = 0x7F
    om = torch.full((G, Mp, N), 3.0, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_masked((aq_dev.to(DEV).view(torch.float8_e4m3fn), asc.to(DEV)), (W.to(DEV), Ws.to(DEV)),
                                                    om, masked.to(DEV), 256, True)
    refm = gemm_ref.grouped_gemm_masked(aq, asc, W, Ws, masked)
    for gi in range(G):
        mm = int(masked[gi])
        if mm:
            assert rel_mae(om[gi, :mm].cpu(), refm[gi, :mm]) < 1e-3
        assert bool((om[gi, mm:] == 3.0).all())


def test_big_tile_contiguous_and_dense_vs_oracle():
    """>= 192 rows per group on average: the 256 x 256 tile kernel (grouped_gemm_fp8_big2.hip) in contiguous mode (128-row
    aligned groups, a group boundary INSIDE a 256-row tile span) and as the dense GEMM (M not a multiple of 256)."""
    import deep_gemm

    counts = [384, 128, 640, 256]
    N, K = 640, 768
    xq, xs, W, Ws, ex = make_group_case(counts, N, K, seed=21)
    M = xq.shape[0]
    m_indices = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts)).to(torch.int32)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((xq.to(DEV), xs.to(DEV)), (W.to(DEV), Ws.to(DEV)), out, m_indices.to(DEV), True)
    ref = gemm_ref.grouped_gemm_contiguous(xq, xs, W, Ws, m_indices)
    assert rel_mae(out.cpu(), ref) < 1e-3
    # dense: one group
    xq1, xs1, W1, Ws1, _ = make_group_case([777], 1000, 512, seed=22)
    out1 = torch.zeros(777, 1000, dtype=torch.bfloat16, device=DEV)
    deep_gemm.gemm_fp8_fp8_bf16_nt((xq1.to(DEV), xs1.to(DEV)), (W1[0].to(DEV), Ws1[0].to(DEV)), out1, True)
    ref1 = gemm_ref.grouped_gemm_offset(xq1, xs1, W1, Ws1, torch.tensor([0, 777], dtype=torch.int32))
    assert rel_mae(out1.cpu(), ref1) < 1e-3


def test_set_num_sms_caps_the_grid_and_runs_beside_mla_on_a_second_stream():
    """deep_gemm.set_num_sms (tbo_executor.py:129-134: save, set, restore) — under a cap the grouped GEMM walks its tiles
    with at most `n` workgroups (same results), and it can share the chip with an MLA decode launched on another stream
    inside ONE hipGraph (LongCat runs attention and the MoE GEMMs of two micro-batches concurrently: longcat_flash.py:417-445)."""
    import deep_gemm
    import flash_mla_fp8 as fm
    from helpers import make_paged_case

    counts = [40, 0, 130, 7, 64, 1, 300, 33]
    N, K = 512, 512
    xq, xs, W, Ws, ex = make_group_case(counts, N, K, seed=77)
    M = xq.shape[0]
    xs_dev = torch.zeros((K // 128, M + 64), dtype=torch.float32, device=DEV).permute(-1, -2)
    xs_dev[:M] = xs.to(DEV)
    args = ((xq.to(DEV), xs_dev[:M]), (W.to(DEV), Ws.to(DEV)))
    ex_dev = ex.to(DEV)
    ref_out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset(*args, ref_out, ex_dev)
    full = deep_gemm.get_num_sms()
    assert full >= 64                       # default: the device's CU count
    deep_gemm.set_num_sms(24)
    try:
        assert deep_gemm.get_num_sms() == 24
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset(*args, out, ex_dev)
        assert torch.equal(out, ref_out)    # the same tiles, walked by 24 workgroups
        # MLA decode on a side stream beside the capped GEMM, captured in one graph
        c = make_paged_case([700, 129, 64], 128, seed=3)
        d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in c.items()}
        pages = c["total_pages"]
        qn, qs, qr = fm.quantize_ckv_per_token_head(d["q"].contiguous(), 512)
        meta, ns = fm.get_mla_metadata(d["cache_seqlens"], 128, 1)

        def mla():
            return fm.flash_mla_ckv_fp8_per_token(qn, qr, d["k_lora"].view(pages, 64, 1, 512), d["k_rope"].view(pages, 64, 1, 64), qs,
                                                  d["k_scale"].view(pages, 64, 1, 1), d["block_table"], d["cache_seqlens"], 512, meta, ns,
                                                  192 ** -0.5, True)

        o_ref, _ = mla()
        torch.cuda.synchronize()
        out2 = torch.zeros_like(out)
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                o_g, _ = mla()
            deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset(*args, out2, ex_dev)
            torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out2, ref_out) and torch.equal(o_g, o_ref)
    finally:
        deep_gemm.set_num_sms(full)         # what tbo_executor does on exit
    assert deep_gemm.get_num_sms() == full
