"""Drop-in `flash_mla_swap` module (reference: 3rdparty/flashmla-swap "SwapAB", imported at
python/sglang/srt/layers/attention/flashmla_backend.py:15; chosen when s_q*H <= 56, :18-22).
On MI355X both modules share one kernel family: the token-major ("swapped") MFMA operand order is
the native layout of the gfx950 kernel for every M."""
from fluent_mi355.mla import flash_mla_with_kvcache, get_mla_metadata  # noqa: F401
