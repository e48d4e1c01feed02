"""flashinfer.norm — the RMSNorm entry points the hot-path files name: `rmsnorm` / `fused_add_rmsnorm` (srt/layers/layernorm.py:26-31, used by
RMSNorm.forward_cuda :64-86) and `_rmsnorm_fused_parallel` (the dual q_a / kv_a norm, :276-303).  One HIP kernel each
(csrc/norm_fused.hip: fl_fused_add_rmsnorm — the arithmetic of RMSNorm.forward_native :88-112, fp32 math, bf16 in / out)."""
import torch

from fluent_mi355.comm import _ops


def _rows(t, what):
    if t.dtype != torch.bfloat16 or not t.is_cuda:
        raise RuntimeError(f"{what}: bf16 CUDA/HIP tensors only (got {t.dtype} on {t.device})")
    return t.reshape(-1, t.shape[-1])


def rmsnorm(input, weight, eps=1e-6, out=None, enable_pdl=False):
    """out = input * rsqrt(mean(input^2) + eps) * weight (rows = the last dimension); `enable_pdl` is accepted and ignored."""
    x = _rows(input, "rmsnorm")
    xc = x if x.is_contiguous() else x.contiguous()
    if out is None:
        out = torch.empty_like(input, memory_format=torch.contiguous_format)
    o = out.reshape(-1, out.shape[-1])
    oc = o if o.is_contiguous() else torch.empty_like(xc)
    _ops().add_rmsnorm(xc.unsqueeze(0), None, None, weight.contiguous(), eps, None, oc, None, None)
    if oc is not o:
        o.copy_(oc)
    return out


def fused_add_rmsnorm(input, residual, weight, eps=1e-6, enable_pdl=False):
    """IN PLACE (flashinfer semantics, layernorm.py:76-79): residual += input; input = rmsnorm(residual)."""
    x, r = _rows(input, "fused_add_rmsnorm"), _rows(residual, "fused_add_rmsnorm")
    if not (x.is_contiguous() and r.is_contiguous()):
        xc, rc = x.contiguous(), r.contiguous()
        _ops().add_rmsnorm(xc.unsqueeze(0), None, rc, weight.contiguous(), eps, rc, xc, None, None)
        x.copy_(xc)
        r.copy_(rc)
        return
    # (every thread reads its elements of both rows before it writes them: the in-place form is safe)
    _ops().add_rmsnorm(x.unsqueeze(0), None, r, weight.contiguous(), eps, r, x, None, None)


def _rmsnorm_fused_parallel(input1, weight1, output1, input2, weight2, output2, eps=1e-6, enable_pdl=False):
    """The two independent norms of the MLA down-projection (q_a [T, q_lora_rank], kv_a [T, kv_lora_rank]; layernorm.py:276-303).  The
    hot path itself runs them inside the gather (C7: flashinfer.comm.trtllm_allgather_fusion); this is the exchange-free form."""
    rmsnorm(input1, weight1, eps, out=output1)
    rmsnorm(input2, weight2, eps, out=output2)


def gemma_rmsnorm(input, weight, eps=1e-6, out=None, enable_pdl=False):
    """Gemma form (layernorm.py:26-31,209-233): the learned weight is an offset from 1."""
    return rmsnorm(input, weight + 1.0, eps, out=out)


def gemma_fused_add_rmsnorm(input, residual, weight, eps=1e-6, enable_pdl=False):
    fused_add_rmsnorm(input, residual, weight + 1.0, eps)
