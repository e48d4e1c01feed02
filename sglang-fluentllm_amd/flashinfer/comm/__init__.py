"""`flashinfer.comm` drop-in: the comm-fused ops of the decode step (python/sglang/srt/layers/flashinfer_comm_fusion.py)."""
from fluent_mi355.comm import (AllGatherFusionPattern, AllReduceFusionPattern, ReduceScatterFusionPattern,  # noqa: F401
                               destroy_ipc_workspace_for_allgather, trtllm_allgather_fusion, trtllm_allreduce_fusion,
                               trtllm_create_ipc_workspace_for_all_reduce_fusion,
                               trtllm_destroy_ipc_workspace_for_all_reduce_fusion, trtllm_reducescatter_fusion)

from . import all_gather, vllm_ar  # noqa: F401
