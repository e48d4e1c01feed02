"""eps.utils.ops — `_ops` is the compiled-op namespace of the reference's eps package (topk.py:47)."""
from . import _ops  # noqa: F401
