"""Drop-in `flash_mla_fp8` module (reference: 3rdparty/flashmla-fp8, imported at
python/sglang/srt/layers/attention/flashmla_backend.py:14 and mem_cache/memory_pool.py:45).
MI355X-native: every function calls hand-written gfx950 HIP kernels through libfluent_mi355.so."""
from fluent_mi355.mla import (  # noqa: F401
    dequantize_ckv_fused_indexed,
    flash_mla_ckv_fp8_per_token,
    flash_mla_with_kvcache,
    get_mla_metadata,
    quantize_and_cache_k,
    quantize_ckv_per_token_head,
    quantize_q_and_cache_k,   # K5 + K4 in one launch: an extension over the reference module
    absorb_rope_quant,        # absorption bmm + RoPE + K5 + K4 in one launch: likewise an extension
    flash_mla_ckv_fp8_per_token_bf16_q,   # K4 inside the decode kernel's prologue: likewise an extension
)
