"""Host-side mirror of `flashinfer.apply_rope_with_cos_sin_cache_inplace` as the reference's RotaryEmbedding.forward_cuda
calls it (python/sglang/srt/layers/rotary_embedding.py:203-218): rotary embedding of query / key in place, one HIP kernel
(csrc/rope.hip), including the two extras the decode path passes (models/deepseek_v2.py:843-861): `output_q_rope` (the
rotated query goes into the rope columns of the absorbed Q instead of in place) and `fused_set_kv_buffer_arg`
(`FusedSetKVBufferArg`, models/utils.py:52-81: bf16 KV cache only — the same launch stores value and rotated key into the
cache rows).  No fallback."""
from __future__ import annotations

import ctypes

import torch

import dataclasses
from typing import Optional

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




class _FlRopeArgs(ctypes.Structure):
    _fields_ = [("positions", _vp), ("num_tokens", _i64),
                ("q", _vp), ("q_stride_token", _i64), ("q_stride_head", _i64), ("num_q_heads", _i),
                ("k", _vp), ("k_stride_token", _i64), ("k_stride_head", _i64), ("num_k_heads", _i),
                ("cos_sin_cache", _vp), ("max_position", _i64), ("rotary_dim", _i), ("is_neox", _i),
                ("q_out", _vp), ("qo_stride_token", _i64), ("qo_stride_head", _i64),
                ("k_out", _vp), ("ko_stride_token", _i64), ("ko_stride_head", _i64),
                ("k_buffer", _vp), ("k_buffer_stride", _i64), ("v_buffer", _vp), ("v_buffer_stride", _i64),
                ("value", _vp), ("value_stride", _i64), ("value_dim", _i), ("cache_loc", _vp), ("cache_loc_is_i64", _i)]


lib.fl_rope.argtypes = [ctypes.POINTER(_FlRopeArgs), _vp]
lib.fl_rope.restype = _i


@dataclasses.dataclass
class FusedSetKVBufferArg:
    """flashinfer.FusedSetKVBufferArg as models/utils.py:73-80 builds it: `value` [T, (1,) D_v] -> v_buffer[cache_loc],
    the rotated key -> k_buffer[cache_loc]; buffers are 2-D row views of the bf16 KV pool (row stride = the pool's)."""
    value: torch.Tensor
    k_buffer: torch.Tensor
    v_buffer: torch.Tensor
    k_scale: Optional[float]
    v_scale: Optional[float]
    cache_loc: torch.Tensor


def apply_rope_with_cos_sin_cache_inplace(positions, query, key, head_size, cos_sin_cache, is_neox=True,
                                          fused_set_kv_buffer_arg=None, output_q_rope=None, output_k_rope=None):
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
    a = _FlRopeArgs()
    a.positions, a.num_tokens = pos.data_ptr(), T
    a.q, a.q_stride_token, a.q_stride_head, a.num_q_heads = query.data_ptr(), qst, qsh, hq
    a.k, a.k_stride_token, a.k_stride_head, a.num_k_heads = key.data_ptr(), kst, ksh, hk
    a.cos_sin_cache, a.max_position, a.rotary_dim, a.is_neox = cache.data_ptr(), cache.shape[0], cache.shape[1], int(bool(is_neox))
    keep = [pos, cache]
    for name, out, heads in (("q", output_q_rope, hq), ("k", output_k_rope, hk)):
        if out is None:
            continue
        To, ho, ost, osh = _rows(out, head_size, f"output_{name}_rope")
        if To != T or ho != heads:
            raise RuntimeError(f"output_{name}_rope must have the token / head counts of its source")
        setattr(a, f"{name}_out", out.data_ptr())
        setattr(a, f"{name}o_stride_token", ost)
        setattr(a, f"{name}o_stride_head", osh)
    f = fused_set_kv_buffer_arg
    if f is not None:
        for sc, nm in ((f.k_scale, "k_scale"), (f.v_scale, "v_scale")):
            if sc is not None and float(sc) != 1.0:
                raise RuntimeError(f"fused set-KV: {nm}={sc}: only the bf16 cache is served (models/utils.py:35-49 disables the "
                                   "fused path for fp8 KV; that cache is written by flash_mla_fp8.quantize_and_cache_k)")
        if cache.shape[1] != head_size:
            raise RuntimeError("fused set-KV: the key rows must be rotary_dim wide (MLA k_pe)")
        kb, vb, val = f.k_buffer, f.v_buffer, f.value
        val2 = val.reshape(T, -1) if val.is_contiguous() else val.view(T, -1)
        for t_, nm in ((kb, "k_buffer"), (vb, "v_buffer"), (val2, "value")):
            if t_.dtype != torch.bfloat16 or t_.dim() != 2 or t_.stride(1) != 1 or not t_.is_cuda:
                raise RuntimeError(f"fused set-KV: {nm} must be a 2-D bf16 CUDA/HIP row view")
        if kb.shape[1] != hk * head_size or vb.shape[1] != val2.shape[1]:
            raise RuntimeError("fused set-KV: buffer widths do not match key / value")
        loc = f.cache_loc.reshape(-1)
        if loc.numel() != T or loc.dtype not in (torch.int64, torch.int32):
            raise RuntimeError("fused set-KV: cache_loc must be int64 / int32 [tokens]")
        loc = loc.contiguous()
        a.k_buffer, a.k_buffer_stride, a.v_buffer, a.v_buffer_stride = kb.data_ptr(), kb.stride(0), vb.data_ptr(), vb.stride(0)
        a.value, a.value_stride, a.value_dim = val2.data_ptr(), val2.stride(0), val2.shape[1]
        a.cache_loc, a.cache_loc_is_i64 = loc.data_ptr(), int(loc.dtype == torch.int64)
        keep += [val2, loc]
    if T == 0:
        return
    check(lib.fl_rope(ctypes.byref(a), stream_ptr(query.device)), "fl_rope")
