"""Drop-in `deep_gemm` module (reference: 3rdparty/deep_gemm "SwapAB Offset + PDL", imported at
python/sglang/srt/layers/moe/gemms/fp8/fire.py:3, moe/executors/deep_ep_executor.py:6-10,
dense/gemms/fp8/deep_geem.py:6, tbo/tbo_executor.py:16).  MI355X-native: gfx950 MFMA kernels behind libfluent_mi355.so."""
from fluent_mi355.gemm import (  # noqa: F401
    ceil_div,
    gemm_fp8_fp8_bf16_nt,
    get_col_major_tma_aligned_tensor,
    get_num_sms,
    m_grouped_gemm_fp8_fp8_bf16_nt_contiguous,
    m_grouped_gemm_fp8_fp8_bf16_nt_masked,
    m_grouped_gemm_fp8_fp8_bf16_nt_offset,
    set_num_sms,
)
