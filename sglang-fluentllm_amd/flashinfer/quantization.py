"""flashinfer.quantization — call site python/sglang/srt/layers/moe/executors/fp8_eps_executor.py:53-55,75-77."""
from fluent_mi355.gemm import quant_1x128  # noqa: F401
