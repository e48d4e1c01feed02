"""MI355X: B1, the hand-written bf16 MFMA GEMM behind the weight-absorption bmm's and the router GEMM (csrc/bmm_bf16.hip),
through the C-ABI with the operand layouts of the reference's call sites (srt/models/deepseek_v2.py:840,886,177-179;
weights k-contiguous as :1632-1633 stores them), against a float32 torch reference of the same product: fp32 accumulation
of exact bf16 products, one rounding to bf16 (the rounding torch.bmm's bf16 result has) — tolerance one bf16 ulp."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ulp_close(got, ref32):
    """|got - ref| <= one bf16 ulp of the reference magnitude (+ a small absolute floor for cancellation)"""
    ref_b = ref32.to(torch.bfloat16).float()
    tol = ref32.abs() * 2.0 ** -7 + 2e-3
    return bool(((got.float() - ref_b).abs() <= tol).all())


@pytest.mark.parametrize("T,H", [(1, 16), (37, 16), (128, 128), (256, 32), (300, 128), (700, 128)])
def test_weight_absorption_bmm_pair_matches_fp32_reference(T, H):
    from fluent_mi355.bmm import bmm
    g = torch.Generator().manual_seed(T + H)
    q = torch.randn(T, H, 192, generator=g).to(torch.bfloat16).to(DEV)              # [T, H, nope 128 | rope 64]
    w = (torch.randn(H, 128 + 128, 512, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    w_kc, w_vc = w.split([128, 128], dim=1)
    w_kc = w_kc.transpose(1, 2).contiguous().transpose(1, 2)                         # deepseek_v2.py:1632: [H, 128, 512], k-contiguous
    w_vc = w_vc.contiguous().transpose(1, 2)                                         # :1633: [H, 512, 128], k-contiguous
    q_nope = q[..., :128]
    Q = torch.full((T, H, 576), 7.0, dtype=torch.bfloat16, device=DEV)
    bmm(q_nope.transpose(0, 1), w_kc, out=Q[..., :512].transpose(0, 1))              # the call of deepseek_v2.py:840
    torch.cuda.synchronize()
    ref = torch.einsum("thk,hkn->thn", q_nope.float(), w_kc.float())
    assert _ulp_close(Q[..., :512], ref)
    assert bool((Q[..., 512:] == 7.0).all())                                         # nothing outside the view is written
    attn = torch.randn(T, H, 512, generator=g).to(torch.bfloat16).to(DEV)
    o = bmm(attn.transpose(0, 1), w_vc)                                              # :886
    out = o.transpose(0, 1).flatten(1, 2)
    ref2 = torch.einsum("thk,hkn->thn", attn.float(), w_vc.float()).flatten(1, 2)
    assert out.shape == (T, H * 128) and _ulp_close(out, ref2)


@pytest.mark.parametrize("T", [1, 33, 256, 1000])
def test_router_gemm_matches_fp32_reference(T):
    import flashinfer
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, 7168, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(256, 7168, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    logits = flashinfer.dsv3_router_gemm(x, w, out_dtype=torch.float32)
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t()
    assert logits.dtype == torch.float32 and logits.shape == (T, 256)
    # f32 output: fp32 accumulation in a different order than the library's — relative 1e-5 of the row's magnitude
    assert float((logits - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-4
    lb = flashinfer.dsv3_router_gemm(x, w, out_dtype=torch.bfloat16)
    assert lb.dtype == torch.bfloat16 and _ulp_close(lb, ref)
    # the <= 16-token fused-A GEMM entry point shares the kernel
    wt = w[:, :1536].contiguous().t()                                                # [K = 1536, N = 256] view, k-contiguous
    y = flashinfer.dsv3_fused_a_gemm(x[:16, :1536].contiguous(), wt)
    assert _ulp_close(y, x[:16, :1536].float() @ wt.float())


def test_bmm_refuses_layouts_it_does_not_serve():
    from fluent_mi355.bmm import bmm
    a = torch.zeros(2, 8, 64, dtype=torch.bfloat16, device=DEV)
    b = torch.zeros(2, 64, 32, dtype=torch.bfloat16, device=DEV)                    # n-contiguous: not the call sites' layout
    with pytest.raises(RuntimeError):
        bmm(a, b)
    with pytest.raises(RuntimeError):
        bmm(a.float(), b.transpose(1, 2).contiguous().transpose(1, 2))
    assert bmm(a[:0], b.transpose(1, 2).contiguous().transpose(1, 2)[:0]).shape == (0, 8, 32)
