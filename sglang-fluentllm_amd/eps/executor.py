"""eps.executor.silu(gate_up bf16 [M, 2I], exclusive_sum, num_tokens_hint) -> bf16 [M, I]
(python/sglang/srt/layers/moe/executors/fp8_eps_executor.py:62)."""
from fluent_mi355.gemm import silu  # noqa: F401
