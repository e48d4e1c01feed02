"""Drop-in subset of Meituan `eps` (reference: 3rdparty/eps): the entry points used by FP8EPSExecutor
(python/sglang/srt/layers/moe/executors/fp8_eps_executor.py:11,62)."""
from . import executor  # noqa: F401
