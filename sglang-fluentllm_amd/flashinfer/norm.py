"""flashinfer.norm — the RMSNorm entry points the hot-path files name: `rmsnorm` / `fused_add_rmsnorm` (srt/layers/layernorm.py:26-31, used by
RMSNorm.forward_cuda :64-86) and `_rmsnorm_fused_parallel` (the dual q_a / kv_a norm, :276-303).  One HIP kernel each
(csrc/norm_fused.hip: fl_fused_add_rmsnorm — the arithmetic of RMSNorm.forward_native :88-112, fp32 math, bf16 in / out)."""
import torch

from fluent_mi355.comm import _ops


def _rows(t, what):
    """[.., H] -> [rows, H] as a VIEW of the caller's storage when it can be one (None otherwise: the caller copies in and out)."""
    if t.dtype != torch.bfloat16 or not t.is_cuda:
        raise RuntimeError(f"{what}: bf16 CUDA/HIP tensors only (got {t.dtype} on {t.device})")
    if t.is_contiguous():
        return t.view(-1, t.shape[-1])
    return None


def rmsnorm(input, weight, eps=1e-6, out=None, enable_pdl=False, _gamma_offset=0.0):
    """out = input * rsqrt(mean(input^2) + eps) * weight (rows = the last dimension); `enable_pdl` is accepted and ignored."""
    x = _rows(input, "rmsnorm")
    xc = x if x is not None else input.contiguous().view(-1, input.shape[-1])
    if out is None:
        out = torch.empty_like(input, memory_format=torch.contiguous_format)
    o = _rows(out, "rmsnorm")
    oc = o if o is not None else torch.empty_like(xc)
    _ops().add_rmsnorm(xc.unsqueeze(0), None, None, weight.contiguous(), eps, None, oc, None, None, gamma_offset=_gamma_offset)
    if o is None:
        out.copy_(oc.view(out.shape))   # (a strided `out`: written through the caller's tensor, never a reshaped copy of it)
    return out


def fused_add_rmsnorm(input, residual, weight, eps=1e-6, enable_pdl=False, _gamma_offset=0.0):
    """IN PLACE (flashinfer semantics, layernorm.py:76-79): residual += input; input = rmsnorm(residual)."""
    x, r = _rows(input, "fused_add_rmsnorm"), _rows(residual, "fused_add_rmsnorm")
    if x is None or r is None:
        xc = input.contiguous().view(-1, input.shape[-1])
        rc = residual.contiguous().view(-1, residual.shape[-1])
        _ops().add_rmsnorm(xc.unsqueeze(0), None, rc, weight.contiguous(), eps, rc, xc, None, None, gamma_offset=_gamma_offset)
        input.copy_(xc.view(input.shape))       # (back into the CALLER's tensors)
        residual.copy_(rc.view(residual.shape))
        return
    # (every thread reads its elements of both rows before it writes them; the kernel's aliasable pointers carry no __restrict__)
    _ops().add_rmsnorm(x.unsqueeze(0), None, r, weight.contiguous(), eps, r, x, None, None, gamma_offset=_gamma_offset)


def gemma_rmsnorm(input, weight, eps=1e-6, out=None, enable_pdl=False):
    """Gemma form (layernorm.py:26-31,209-233): x * rsqrt(..) * (1 + w) with the 1 added in fp32 inside the kernel, as the reference does
    (bf16(w + 1) would quantise small gammas to the 2^-7 grid around 1)."""
    return rmsnorm(input, weight, eps, out=out, _gamma_offset=1.0)


def gemma_fused_add_rmsnorm(input, residual, weight, eps=1e-6, enable_pdl=False):
    fused_add_rmsnorm(input, residual, weight, eps, _gamma_offset=1.0)


def _rmsnorm_fused_parallel(input1, weight1, output1, input2, weight2, output2, eps=1e-6, enable_pdl=False):
    """The two independent norms of the MLA down-projection (q_a [T, q_lora_rank], kv_a [T, kv_lora_rank]; layernorm.py:276-303).  The
    hot path itself runs them inside the gather (C7: flashinfer.comm.trtllm_allgather_fusion); this is the exchange-free form."""
    rmsnorm(input1, weight1, eps, out=output1)
    rmsnorm(input2, weight2, eps, out=output2)

