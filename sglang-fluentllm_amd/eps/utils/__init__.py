"""eps.utils — the attribute chain srt/layers/moe/topk.py:46-47 resolves at import time: `utils.ops._ops.topk_sigmoid`."""
from . import ops  # noqa: F401
