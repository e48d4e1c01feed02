"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (torch-CPU) of the FP8 block-scaled GEMM / grouped-GEMM / MoE arithmetic on
the reference's hot path.  The CUDA arithmetic itself lives in the un-vendored `deep_gemm`
submodule (meituan-longcat/DeepGEMM @ feature/swap_ab, empty dir in the reference tree, no
SHA); the math is pinned by the reference's own torch-native test oracles, which these
functions restate and are checked against (tests/golden/gemm_*.npz, produced by
oracle/gen_golden.py running the reference's functions in the build container):

  native_per_token_group_quant_fp8   python/sglang/test/test_block_fp8.py:15-40
  native_w8a8_block_fp8_matmul       python/sglang/test/test_block_fp8.py:89-141
  torch_w8a8_block_fp8_moe           python/sglang/test/test_block_fp8.py:212-241
  SiluAndMul.forward_native          python/sglang/srt/layers/activation.py:58-60
"""
from __future__ import annotations

import torch

FP8_MAX = 448.0
BLOCK = 128


def per_token_group_quant_fp8(x, group_size: int = BLOCK, eps: float = 1e-10):
    """test_block_fp8.py:15-40 — s = max(amax, eps)/448 (fp32), q = clamp(x/s) -> e4m3fn.
    Returns (x_q fp8 [.., K], x_s f32 [.., K/group])."""
    assert x.shape[-1] % group_size == 0 and x.is_contiguous()
    x_ = x.reshape(x.numel() // group_size, group_size)
    amax = x_.abs().max(dim=-1, keepdim=True)[0].clamp(min=eps).to(torch.float32)
    x_s = amax / FP8_MAX
    x_q = (x_ / x_s).clamp(min=-FP8_MAX, max=FP8_MAX).to(torch.float8_e4m3fn)
    return x_q.reshape(x.shape), x_s.reshape(x.shape[:-1] + (x.shape[-1] // group_size,))


def block_fp8_matmul(A, B, As, Bs, block_size=(BLOCK, BLOCK), output_dtype=torch.bfloat16):
    """test_block_fp8.py:89-141 — C = sum_k (A_k @ B_jk^T) * (As[:,k] * Bs[j,k]), fp32 acc.
    A [M,K] fp8, As [M,K/bk] f32; B [N,K] fp8, Bs [N/bn, K/bk] f32."""
    A = A.to(torch.float32)
    B = B.to(torch.float32)
    block_n, block_k = block_size
    M = A.numel() // A.shape[-1]
    N, K = B.shape
    origin = A.shape[:-1] + (N,)
    A = A.reshape(M, K)
    As = As.reshape(M, -1)
    n_tiles = (N + block_n - 1) // block_n
    k_tiles = (K + block_k - 1) // block_k
    assert Bs.shape == (n_tiles, k_tiles) and As.shape[-1] == k_tiles
    C = torch.zeros((M, N), dtype=torch.float32)
    for i in range(k_tiles):
        a = A[:, i * block_k:(i + 1) * block_k]
        for j in range(n_tiles):
            b = B[j * block_n:(j + 1) * block_n, i * block_k:(i + 1) * block_k]
            C[:, j * block_n:(j + 1) * block_n] += torch.matmul(a, b.t()) * (As[:, i:i + 1] * Bs[j, i])
    return C.reshape(origin).to(output_dtype)


def silu_and_mul(x):
    """activation.py:58-60 — silu(x[..., :d]) * x[..., d:] (computed in the input dtype like
    the reference: F.silu on the bf16 tensor, then a bf16 multiply)."""
    d = x.shape[-1] // 2
    return torch.nn.functional.silu(x[..., :d]) * x[..., d:]


# ---- grouped GEMM variants (call sites: moe/gemms/fp8/fire.py:18; deep_ep_executor.py:583,655) ----
def grouped_gemm_offset(A, As, W, Ws, exclusive_sum, output_dtype=torch.bfloat16):
    """m_grouped_gemm_fp8_fp8_bf16_nt_offset: rows [ex[e], ex[e+1]) use expert e.
    A [M,K] fp8, As [M,K/128]; W [E,N,K] fp8, Ws [E,N/128,K/128]. Rows >= ex[E] are untouched (0)."""
    M = A.shape[0]
    out = torch.zeros((M, W.shape[1]), dtype=output_dtype)
    ex = [int(v) for v in exclusive_sum]
    for e in range(W.shape[0]):
        lo, hi = ex[e], ex[e + 1]
        if hi > lo:
            out[lo:hi] = block_fp8_matmul(A[lo:hi], W[e], As[lo:hi], Ws[e], output_dtype=output_dtype)
    return out


def grouped_gemm_contiguous(A, As, W, Ws, m_indices, output_dtype=torch.bfloat16):
    """m_grouped_gemm_fp8_fp8_bf16_nt_contiguous: row i uses expert m_indices[i] (<0 = skip)."""
    M = A.shape[0]
    out = torch.zeros((M, W.shape[1]), dtype=output_dtype)
    for e in range(W.shape[0]):
        mask = m_indices == e
        if mask.any():
            out[mask] = block_fp8_matmul(A[mask], W[e], As[mask], Ws[e], output_dtype=output_dtype)
    return out


def grouped_gemm_masked(A, As, W, Ws, masked_m, output_dtype=torch.bfloat16):
    """m_grouped_gemm_fp8_fp8_bf16_nt_masked: A [G,Mp,K]; only first masked_m[g] rows valid."""
    G, Mp, K = A.shape
    out = torch.zeros((G, Mp, W.shape[1]), dtype=output_dtype)
    for g in range(G):
        m = int(masked_m[g])
        if m:
            out[g, :m] = block_fp8_matmul(A[g, :m], W[g], As[g, :m], Ws[g], output_dtype=output_dtype)
    return out


def moe_fp8_block(a, w1, w2, w1_s, w2_s, topk_weight, topk_ids, block_shape=(BLOCK, BLOCK)):
    """test_block_fp8.py:212-241 with routing given (topk done by the caller):
    quant -> w13 GEMM -> silu·mul -> quant -> w2 GEMM -> weighted sum. a [B,D] bf16."""
    B, D = a.shape
    topk = topk_ids.shape[1]
    a_rep = a.view(B, -1, D).repeat(1, topk, 1).reshape(-1, D)
    out = torch.zeros(B * topk, w2.shape[1], dtype=a.dtype)
    ids = topk_ids.reshape(-1)
    a_q, a_s = per_token_group_quant_fp8(a_rep, block_shape[1])
    a_q = a_q.to(torch.float32)
    for i in range(w1.shape[0]):
        mask = ids == i
        if mask.sum():
            inter = block_fp8_matmul(a_q[mask], w1[i], a_s[mask], w1_s[i], block_shape, output_dtype=a.dtype)
            act = silu_and_mul(inter)
            act_q, act_s = per_token_group_quant_fp8(act, block_shape[1])
            out[mask] = block_fp8_matmul(act_q, w2[i], act_s, w2_s[i], block_shape, output_dtype=a.dtype)
    return (out.view(B, -1, w2.shape[1]) * topk_weight.view(B, -1, 1).to(out.dtype)).sum(dim=1)
