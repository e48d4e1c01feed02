"""Host-side mirror of `flashinfer.apply_rope_with_cos_sin_cache_inplace` as the reference's RotaryEmbedding.forward_cuda
calls it (python/sglang/srt/layers/rotary_embedding.py:203-218): rotary embedding of query / key in place, one HIP kernel
(csrc/rope.hip).  No fallback."""
from __future__ import annotations

import ctypes

import torch

from ._lib import check, lib, stream_ptr

_i64, _vp, _i = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
lib.fl_rope_inplace.argtypes = [_vp, _i64, _vp, _i64, _i64, _i, _vp, _i64, _i64, _i, _vp, _i64, _i, _i, _vp]
lib.fl_rope_inplace.restype = _i


def _rows(x, head_size, name):
    """[T, H*head_size] or [T, H, head_size] (possibly a strided view, e.g. q[..., 128:]) -> (T, H, stride_t, stride_h)"""
    if x.dtype != torch.bfloat16 or not x.is_cuda:
        raise RuntimeError(f"apply_rope_with_cos_sin_cache_inplace: {name} must be a bf16 CUDA/HIP tensor")
    if x.dim() == 2:
        if not x.stride(1) == 1 or x.shape[1] % head_size:
            raise RuntimeError(f"{name}: [T, H*head_size] needs a contiguous last dimension")
        return x.shape[0], x.shape[1] // head_size, x.stride(0), head_size
    if x.dim() == 3 and x.shape[2] == head_size and x.stride(2) == 1:
        return x.shape[0], x.shape[1], x.stride(0), x.stride(1)
    raise RuntimeError(f"{name}: expected [T, H*head_size] or [T, H, head_size] with a contiguous last dimension")


def apply_rope_with_cos_sin_cache_inplace(positions, query, key, head_size, cos_sin_cache, is_neox=True,
                                          fused_set_kv_buffer_arg=None, output_q_rope=None, output_k_rope=None):
    if fused_set_kv_buffer_arg is not None or output_q_rope is not None or output_k_rope is not None:
        raise NotImplementedError("apply_rope_with_cos_sin_cache_inplace: fused set-KV / separate outputs are the bf16-cache "
                                  "path (models/utils.py:35-49 disables them for fp8 KV); not on the FP8 MLA path")
    if cos_sin_cache.dtype != torch.float32:
        raise RuntimeError("cos_sin_cache must be float32 (rotary_embedding.py:113-115 keeps it in fp32 on the GPU path)")
    T, hq, qst, qsh = _rows(query, head_size, "query")
    Tk, hk, kst, ksh = _rows(key, head_size, "key")
    pos = positions.reshape(-1).to(torch.int64).contiguous()
    if Tk != T or pos.numel() != T:
        raise RuntimeError("positions / query / key disagree on the number of tokens")
    cache = cos_sin_cache.contiguous()
    if cache.dim() != 2 or cache.shape[1] > head_size or cache.shape[1] % 2:
        raise RuntimeError(f"cos_sin_cache must be [max_position, rotary_dim] with an even rotary_dim <= head_size "
                           f"(got {tuple(cache.shape)}, head_size {head_size})")
    if T == 0:
        return
    check(lib.fl_rope_inplace(pos.data_ptr(), T, query.data_ptr(), qst, qsh, hq, key.data_ptr(), kst, ksh, hk, cache.data_ptr(),
                              cache.shape[0], cache.shape[1], int(bool(is_neox)), stream_ptr(query.device)), "fl_rope_inplace")
