"""eps.fast_ep.AllToAll — call site python/sglang/srt/layers/moe/dispatcher/fast_ep.py:2,16-22."""
from fluent_mi355.ep import AllToAll  # noqa: F401
