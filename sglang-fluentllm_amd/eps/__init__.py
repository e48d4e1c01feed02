"""Drop-in subset of Meituan `eps` (reference: 3rdparty/eps): the entry points used by FP8EPSExecutor and FastEP
(python/sglang/srt/layers/moe/executors/fp8_eps_executor.py:11,62; moe/dispatcher/fast_ep.py:2)."""
from . import communication, executor, fast_ep, utils  # noqa: F401
