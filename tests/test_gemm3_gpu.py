"""G1 / G3 / G4 on the round-6 192 x 256 one-wave-per-SIMD tile (csrc/grouped_gemm_fp8_big3.hip): HIP vs oracle.gemm_ref on the same seeded
inputs (rel-MAE < 1e-3: the reference's own block-fp8 threshold), rows past a group's end untouched.  Shapes chosen so the dispatch takes this
kernel (>= 128 rows per group, N % 256 == 0, K <= 8192): ragged row counts around the 192-row tile edge, empty groups, several n tiles, the two
weight-scale rows of a tile, K with an odd number of k blocks (the ring position of a tile's first stage alternates), tiles carried across the
persistent walk (more tiles than CUs) and masked / dense modes."""
import numpy as np
import pytest
import torch

from oracle import gemm_ref
from test_gemm_gpu import DEV, fp8_weights, make_group_case, rel_mae

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("counts,N,K", [([192, 193, 1, 0, 383, 385, 200], 256, 256),
                                        ([512, 130, 700], 512, 384),
                                        ([193, 225, 257, 289, 321, 353, 32, 64, 96, 128, 160], 256, 512),   # last tiles with 1..6 blocks of 32 rows: skipped MFMA groups
                                        ([191, 577, 0, 256], 768, 640),
                                        ([300] * 40, 1024, 256),            # 80 m tiles x 4 n tiles = 320 tiles > 256 CUs: carried tiles
                                        ([530, 490, 512, 600], 4096, 7168),   # BASELINE config 3 w13 shape, rows around 512
                                        ([400, 128], 7168, 2048)])            # w2 shape
def test_big3_offset_vs_oracle(counts, N, K):
    import deep_gemm

    xq, xs, W, Ws, ex = make_group_case(counts, N, K, seed=len(counts) * 31 + N)
    M, E = xq.shape[0], len(counts)
    mp = (M + E * 31) // 32 * 32 + 32
    xs_dev = torch.zeros((K // 128, mp), dtype=torch.float32, device=DEV).permute(-1, -2)   # the executor's column-major scales
    xs_dev[:M] = xs.to(DEV)
    out = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq.to(DEV), xs_dev[:M]), (W.to(DEV), Ws.to(DEV)), out[:M], ex.to(DEV), use_pdl=True)
    torch.cuda.synchronize()
    ref = gemm_ref.grouped_gemm_offset(xq, xs, W, Ws, ex)
    o = out[:M].cpu()
    assert torch.isfinite(o.float()).all()
    # per group, so that a wrong small group cannot hide behind a large one
    for e in range(E):
        a, b = int(ex[e]), int(ex[e + 1])
        if b > a:
            assert rel_mae(o[a:b], ref[a:b]) < 1e-3, (e, a, b)
    assert bool((out[M:] == 7.0).all())


def test_big3_masked_and_dense_vs_oracle():
    import deep_gemm

    G, Mp, N, K = 4, 576, 512, 512
    masked = torch.tensor([576, 0, 193, 384], dtype=torch.int32)
    g = torch.Generator().manual_seed(15)
    xm = (torch.randn(G, Mp, K, generator=g) / 3).to(torch.bfloat16)
    aq, asc = gemm_ref.per_token_group_quant_fp8(xm, 128)
    W = fp8_weights(g, G, N, K)
    Ws = torch.rand(G, N // 128, K // 128, generator=g) * 1e-2
    aq_dev = aq.clone().view(torch.uint8)
    for gi in range(G):
        aq_dev[gi, int(masked[gi]):] = 0x7F
    om = torch.full((G, Mp, N), 3.0, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_masked((aq_dev.to(DEV).view(torch.float8_e4m3fn), asc.to(DEV)), (W.to(DEV), Ws.to(DEV)),
                                                    om, masked.to(DEV), 400, True)
    refm = gemm_ref.grouped_gemm_masked(aq, asc, W, Ws, masked)
    for gi in range(G):
        mm = int(masked[gi])
        if mm:
            assert rel_mae(om[gi, :mm].cpu(), refm[gi, :mm]) < 1e-3, gi
        assert bool((om[gi, mm:] == 3.0).all())
    # dense: one group, M not a multiple of 192
    xq1, xs1, W1, Ws1, _ = make_group_case([777], 1024, 512, seed=23)
    out1 = torch.zeros(777, 1024, dtype=torch.bfloat16, device=DEV)
    deep_gemm.gemm_fp8_fp8_bf16_nt((xq1.to(DEV), xs1.to(DEV)), (W1[0].to(DEV), Ws1[0].to(DEV)), out1, True)
    ref1 = gemm_ref.grouped_gemm_offset(xq1, xs1, W1, Ws1, torch.tensor([0, 777], dtype=torch.int32))
    assert rel_mae(out1.cpu(), ref1) < 1e-3


def test_big3_matches_big2_bitwise_scale_handling_on_extreme_scales(monkeypatch):
    """Zero, denormal and huge token scales (the E8M0 / mantissa split of the block scale): finite, and equal to the oracle."""
    import deep_gemm

    counts, N, K = [260, 200], 256, 512
    xq, xs, W, Ws, ex = make_group_case(counts, N, K, seed=77)
    xs = xs.clone()
    xs[3, 1] = 0.0
    xs[5, 0] = 1e-41        # denormal
    xs[7, 2] = 3e4
    xs[300, 3] = 2.0 ** -100
    M = xq.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq.to(DEV), xs.to(DEV)), (W.to(DEV), Ws.to(DEV)), out, ex.to(DEV), use_pdl=True)
    ref = gemm_ref.grouped_gemm_offset(xq, xs, W, Ws, ex)
    o = out.cpu()
    assert torch.isfinite(o.float()).all()
    assert rel_mae(o, ref) < 1e-3
