"""flashinfer.activation — call sites python/sglang/srt/layers/activation.py:73,
python/sglang/srt/layers/moe/executors/deep_ep_executor.py:676."""
from fluent_mi355.gemm import silu_and_mul, silu_and_mul_fuse_block_quant  # noqa: F401
