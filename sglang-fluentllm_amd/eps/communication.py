"""eps.communication.TPDPConvertor — call site python/sglang/srt/layers/dp_attention.py:62-74."""
from fluent_mi355.comm import TPDPConvertor  # noqa: F401
