"""eps.utils.ops._ops — the one op the hot-path files name (srt/layers/moe/topk.py:47)."""
from fluent_mi355.router import topk_sigmoid  # noqa: F401
