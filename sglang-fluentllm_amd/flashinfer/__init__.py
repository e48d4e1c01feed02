"""Drop-in subset of `flashinfer` (reference fork: 3rdparty/flashinfer feature/longcat_main): ONLY the entry points on
the FP8 MLA-decode / grouped-GEMM hot path: quantization.quant_1x128, sgl_per_token_group_quant_fp8,
silu_and_mul, activation.silu_and_mul_fuse_block_quant, comm.trtllm_{allreduce,reducescatter,allgather}_fusion, and
moe_fused_gate (the router selection feeding the EP dispatch, srt/layers/moe/topk.py:33,715) and
apply_rope_with_cos_sin_cache_inplace (q_pe / k_pe rotation before K5/K4, srt/layers/rotary_embedding.py:31,203), plus the
rmsnorm / fused_add_rmsnorm / norm._rmsnorm_fused_parallel (layernorm.py:26-31,293), topk_softmax / routing_flash (topk.py:33,845), the
two attention-wrapper classes FlashMLABackend's base class constructs (flashinfer_mla_backend.py:124-142) as INERT objects
(attention_wrappers.py) and `comm.vllm_ar` (C4).  Sampling / norm / prefill attention are out of scope (not named by the
north star)."""
from fluent_mi355.gemm import sgl_per_token_group_quant_fp8, silu_and_mul  # noqa: F401
from fluent_mi355.router import moe_fused_gate, routing_flash, topk_softmax  # noqa: F401
from fluent_mi355.rope import FusedSetKVBufferArg, apply_rope_with_cos_sin_cache_inplace  # noqa: F401

from .attention_wrappers import BatchMLAPagedAttentionWrapper, BatchPrefillWithRaggedKVCacheWrapper  # noqa: F401

from . import activation, comm, norm, quantization  # noqa: F401
from .norm import fused_add_rmsnorm, gemma_fused_add_rmsnorm, gemma_rmsnorm, rmsnorm  # noqa: F401

import torch as _torch


from fluent_mi355.bmm import dsv3_router_gemm  # noqa: F401,E402  (hand-written bf16 MFMA GEMM, csrc/bmm_bf16.hip)


def dsv3_fused_a_gemm(hidden_states, weight_t):
    """models/deepseek_v2.py:46,785-793: the <= 16-token min-latency q_a/kv_a projection, behind
    `use_min_latency_fused_a_gemm` (an sm90 switch): hidden [T, K] x weight_t [K, N] with weight_t k-contiguous (the
    transposed view of the [N, K] parameter, as the call site passes it) -> the same hand-written MFMA kernel."""
    from fluent_mi355.bmm import bmm as _bmm

    return _bmm(hidden_states.unsqueeze(0), weight_t.unsqueeze(0))[0]


def merge_state(v_a, s_a, v_b, s_b):
    """models/deepseek_v2.py:43 imports this name for the chunked-PREFILL attention merge (out of the decode hot path,
    SURVEY §8 'out of scope').  Importable; calling it is refused."""
    raise RuntimeError("flashinfer.merge_state: prefill attention state merge is outside the MI355X hot-path build")
