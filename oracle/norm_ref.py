"""ORACLE (test infrastructure only — never imported by the product path): CPU restatement of the norm / comm-fusion
arithmetic on the decode path.

  rmsnorm_native     <- RMSNorm.forward_native, /root/reference/python/sglang/srt/layers/layernorm.py:88-112
  fused_add_rmsnorm  <- what flashinfer.comm.trtllm_{allreduce,reducescatter}_fusion compute after the exchange
                        (call sites flashinfer_comm_fusion.py:372-397, 485-509 <- layernorm.py:114-189): fp32 sum of the
                        ranks' pieces (+ add_in) + residual, then RMSNorm, then the optional 1x128 FP8 block quantisation
                        of the bf16 norm (test/test_block_fp8.py:15-40 via oracle.gemm_ref)
  dual_rmsnorm       <- trtllm_allgather_fusion's dual norm (flashinfer_comm_fusion.py:613-638 <- layernorm.py:305-359)
Pinned by tests/golden/rmsnorm_native.npz (outputs of the REAL reference method, oracle/gen_golden.py)."""
import torch


def rmsnorm_native(x, weight, eps, residual=None):
    orig = x.dtype
    x = x.to(torch.float32)
    if residual is not None:
        x = x + residual.to(torch.float32)
        residual = x.to(orig)
    var = x.pow(2).mean(dim=-1, keepdim=True)
    x = x * torch.rsqrt(var + eps)
    x = (x * weight).to(orig)
    return x if residual is None else (x, residual)


def fused_add_rmsnorm(pieces, add_in, residual_in, gamma, eps):
    """pieces [W, T, H] bf16 -> (norm bf16, residual_out bf16)."""
    v = pieces.float().sum(0)
    if add_in is not None:
        v = v + add_in.float()
    if residual_in is not None:
        v = v + residual_in.float()
    res = v.to(pieces.dtype)
    var = v.pow(2).mean(dim=-1, keepdim=True)
    y = (v * torch.rsqrt(var + eps) * gamma.float()).to(pieces.dtype)
    return y, res


def dual_rmsnorm(ag, q_rank, kv_rank, gamma_q, gamma_kv, eps_q, eps_kv):
    """ag [T, D] -> (x_norm [T, q_rank], ag with cols [q_rank, q_rank+kv_rank) normalised)."""
    x = rmsnorm_native(ag[:, :q_rank], gamma_q, eps_q)
    out = ag.clone()
    out[:, q_rank:q_rank + kv_rank] = rmsnorm_native(ag[:, q_rank:q_rank + kv_rank], gamma_kv, eps_kv)
    return x, out
