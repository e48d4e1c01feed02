"""TEST-ONLY torch implementation of the two fused kernels behind fluent_mi355.comm (csrc/norm_fused.hip), so that the
multi-process HOST logic (one-shot exchanges with uneven token splits, residual scattering, output slicing) can run on CPU
tensors under the gloo backend.  Uses the oracle's arithmetic; never used by the product path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gemm_ref, norm_ref  # noqa: E402


class TorchNormOps:
    def add_rmsnorm(self, pieces, add_in, residual_in, gamma, eps, residual_out, norm_out, quant_out, scale_out):
        if gamma is None:   # sum-only (one-shot reduce-scatter)
            residual_out.copy_(pieces.float().sum(0).to(pieces.dtype))
            return
        y, res = norm_ref.fused_add_rmsnorm(pieces, add_in, residual_in, gamma, eps)
        if residual_out is not None:
            residual_out.copy_(res)
        if norm_out is not None:
            norm_out.copy_(y)
        if quant_out is not None:
            q, s = gemm_ref.per_token_group_quant_fp8(y.contiguous(), 128)
            quant_out.copy_(q)
            scale_out.copy_(s)

    def dual_rmsnorm(self, ag, q_rank, kv_rank, gamma_q, gamma_kv, eps_q, eps_kv, x_norm_out, quant_out, scale_out):
        x, out = norm_ref.dual_rmsnorm(ag, q_rank, kv_rank, gamma_q, gamma_kv, eps_q, eps_kv)
        ag.copy_(out)
        if x_norm_out is not None:
            x_norm_out.copy_(x)
        if quant_out is not None:
            q, s = gemm_ref.per_token_group_quant_fp8(x.contiguous(), 128)
            quant_out.copy_(q)
            scale_out.copy_(s)
