"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy float32) of the reference's rotary embedding on the MLA decode path:
`DeepseekScalingRotaryEmbedding.forward_native` (python/sglang/srt/layers/rotary_embedding.py:804-846) applied to
q_pe / k_pe (models/deepseek_v2.py:646-647,695-696) with the fp32 cos/sin cache the CUDA path keeps (:113-115) — the math of
`flashinfer.apply_rope_with_cos_sin_cache_inplace` (:203-218).  Pinned against tests/golden/rope_deepseek_yarn.npz.
Only tests/ may import this."""
import numpy as np


def bf16_to_f32(u16):
    return (np.asarray(u16, np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)          # round to nearest even (no NaN inputs in tests)
    return r


def apply_rope(positions, x_u16, cos_sin_cache, is_neox):
    """x_u16: bf16 bits [T, H, D]; cos_sin_cache f32 [max_pos, R] (cos | sin halves, R = rotary_dim <= D).
    -> bf16 bits [T, H, D].  fp32 math with separate roundings of the two products and of their sum (torch eager)."""
    x = bf16_to_f32(x_u16)
    T, H, D = x.shape
    cache = np.asarray(cos_sin_cache, np.float32)
    R = cache.shape[1]
    cs = cache[np.asarray(positions, np.int64)]                            # :822-824
    cos, sin = cs[:, None, : R // 2], cs[:, None, R // 2:]                  # :825
    out = x.copy()
    rot = x[..., :R]
    if is_neox:                                                           # :826-830, _rotate_neox :50-53
        x1, x2 = rot[..., : R // 2], rot[..., R // 2:]
        o1 = (x1 * cos).astype(np.float32) + ((-x2) * sin).astype(np.float32)
        o2 = (x2 * cos).astype(np.float32) + (x1 * sin).astype(np.float32)
        out[..., : R // 2], out[..., R // 2: R] = o1, o2
    else:                                                                 # :831-833, _rotate_gptj :56-60
        x1, x2 = rot[..., 0::2], rot[..., 1::2]
        out[..., 0:R:2] = (x1 * cos).astype(np.float32) + ((-x2) * sin).astype(np.float32)
        out[..., 1:R:2] = (x2 * cos).astype(np.float32) + (x1 * sin).astype(np.float32)
    return f32_to_bf16(out)                                               # :846 .to(dtype)
