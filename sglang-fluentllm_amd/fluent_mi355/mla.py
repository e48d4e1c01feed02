"""Host-side mirror of the `flash_mla_fp8` / `flash_mla_swap` Python operator API.

Same names, keyword arguments and return conventions as the calls the reference's
FlashMLABackend makes (python/sglang/srt/layers/attention/flashmla_backend.py:125-175,206-254,
261-265; python/sglang/srt/mem_cache/memory_pool.py:821-824,864-871).  Every function marshals
`tensor.data_ptr()` + the current HIP stream into the C-ABI (include/fluent_mi355.h); outputs are
allocated with torch.empty on the input device (so the caching allocator / graph pool owns
them), caches are mutated in place, failures raise RuntimeError.  Nothing here computes.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from ._lib import FlMlaDecodeArgs, check, cu_count, lib, stream_ptr

PAGE_SIZE = 64
META_W = 8
KV_FP8_PER_TOKEN, KV_FP8_576, KV_BF16_576 = 0, 1, 2
_ONE_BYTE = (torch.uint8, torch.int8, torch.float8_e4m3fn)


def _req(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _cuda_contig(t: torch.Tensor, name: str):
    _req(t.is_cuda, f"{name} must be a device tensor (no CPU fallback on this path)")
    _req(t.is_contiguous(), f"{name} must be contiguous")


def get_mla_metadata(cache_seqlens: torch.Tensor, num_heads_per_head_k: int, num_heads_k: int = 1
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (tile_scheduler_metadata int32 [num_parts, 8], num_splits int32 [bs+1]).
    Shapes depend only on (device CU count, num_heads_per_head_k) and bs, as the reference's persistent
    graph buffers require (flashmla_backend.py:307-321,340-341). Runs on-device, no host sync."""
    _req(num_heads_k == 1, "MLA has one latent KV head (num_heads_k must be 1)")
    _req(cache_seqlens.dtype == torch.int32, "cache_seqlens must be int32")
    _cuda_contig(cache_seqlens, "cache_seqlens")
    bs = cache_seqlens.shape[0]
    dev = cache_seqlens.device
    num_parts = lib.fl_mla_num_parts(cu_count(dev), int(num_heads_per_head_k))
    meta = torch.empty((num_parts, META_W), dtype=torch.int32, device=dev)
    num_splits = torch.empty((bs + 1,), dtype=torch.int32, device=dev)
    check(lib.fl_mla_get_metadata(cache_seqlens.data_ptr(), bs, num_parts, meta.data_ptr(), num_splits.data_ptr(),
                                  stream_ptr(dev)), "fl_mla_get_metadata")
    return meta, num_splits


def quantize_ckv_per_token_head(q: torch.Tensor, kv_lora_rank: int = 512):
    """q bf16 [bs, s_q, H, 576] -> (q_nope fp8 [..,512], q_scale f32 [..,1], q_rope bf16 [..,64])."""
    _cuda_contig(q, "q")
    _req(q.dtype == torch.bfloat16, "q must be bfloat16")
    d = q.shape[-1]
    d_rope = d - kv_lora_rank
    rows = q.numel() // d
    q_nope = torch.empty(q.shape[:-1] + (kv_lora_rank,), dtype=torch.float8_e4m3fn, device=q.device)
    q_scale = torch.empty(q.shape[:-1] + (1,), dtype=torch.float32, device=q.device)
    q_rope = torch.empty(q.shape[:-1] + (d_rope,), dtype=torch.bfloat16, device=q.device)
    check(lib.fl_mla_quant_q(q.data_ptr(), rows, kv_lora_rank, d_rope, q_nope.data_ptr(), q_scale.data_ptr(),
                             q_rope.data_ptr(), stream_ptr(q.device)), "fl_mla_quant_q")
    return q_nope, q_scale, q_rope


def quantize_and_cache_k(key: torch.Tensor, k_lora_cache: torch.Tensor, k_lora_scale_cache: torch.Tensor,
                         k_rope_cache: torch.Tensor, indices: torch.Tensor, head_dim_v: int = 512) -> None:
    """In place: per-token quantise key [n,1,576] and scatter into the three caches at `indices` (int32)."""
    for t, n in ((key, "key"), (k_lora_cache, "k_lora_cache"), (k_lora_scale_cache, "k_lora_scale_cache"),
                 (k_rope_cache, "k_rope_cache"), (indices, "indices")):
        _cuda_contig(t, n)
    _req(key.dtype == torch.bfloat16 and k_rope_cache.dtype == torch.bfloat16, "key / rope cache must be bfloat16")
    _req(k_lora_cache.dtype in _ONE_BYTE and k_lora_scale_cache.dtype == torch.float32, "bad cache dtypes")
    _req(indices.dtype == torch.int32, "indices must be int32")
    d = key.shape[-1]
    n = key.numel() // d
    _req(indices.numel() == n, "indices / key length mismatch")
    num_slots = k_lora_cache.numel() // head_dim_v
    check(lib.fl_mla_quant_store_k(key.data_ptr(), n, head_dim_v, d - head_dim_v, indices.data_ptr(),
                                   k_lora_cache.data_ptr(), k_lora_scale_cache.data_ptr(), k_rope_cache.data_ptr(),
                                   num_slots, stream_ptr(key.device)), "fl_mla_quant_store_k")


def quantize_q_and_cache_k(q: torch.Tensor, key: torch.Tensor, k_lora_cache: torch.Tensor, k_lora_scale_cache: torch.Tensor,
                           k_rope_cache: torch.Tensor, indices: torch.Tensor, kv_lora_rank: int = 512):
    """`quantize_and_cache_k(key, ...)` and `quantize_ckv_per_token_head(q, kv_lora_rank)` in ONE launch — the two calls
    FlashMLABackend.forward_decode issues back to back (flashmla_backend.py:188-206).  Same bytes as the separate calls;
    returns (q_nope, q_scale, q_rope).  Optional entry point for an integrator: the unmodified
    backend keeps working with the two separate functions."""
    for t, n in ((q, "q"), (key, "key"), (k_lora_cache, "k_lora_cache"), (k_lora_scale_cache, "k_lora_scale_cache"),
                 (k_rope_cache, "k_rope_cache"), (indices, "indices")):
        _cuda_contig(t, n)
    _req(q.dtype == torch.bfloat16 and key.dtype == torch.bfloat16 and k_rope_cache.dtype == torch.bfloat16,
         "q / key / rope cache must be bfloat16")
    _req(k_lora_cache.dtype in _ONE_BYTE and k_lora_scale_cache.dtype == torch.float32, "bad cache dtypes")
    _req(indices.dtype == torch.int32, "indices must be int32")
    d = q.shape[-1]
    _req(key.shape[-1] == d, "q / key row width mismatch")
    d_rope = d - kv_lora_rank
    n_k = key.numel() // d
    _req(indices.numel() == n_k, "indices / key length mismatch")
    rows = q.numel() // d
    q_nope = torch.empty(q.shape[:-1] + (kv_lora_rank,), dtype=torch.float8_e4m3fn, device=q.device)
    q_scale = torch.empty(q.shape[:-1] + (1,), dtype=torch.float32, device=q.device)
    q_rope = torch.empty(q.shape[:-1] + (d_rope,), dtype=torch.bfloat16, device=q.device)
    num_slots = k_lora_cache.numel() // kv_lora_rank
    check(lib.fl_mla_quant_q_store_k(key.data_ptr(), n_k, indices.data_ptr(), k_lora_cache.data_ptr(),
                                     k_lora_scale_cache.data_ptr(), k_rope_cache.data_ptr(), num_slots, q.data_ptr(), rows,
                                     kv_lora_rank, d_rope, q_nope.data_ptr(), q_scale.data_ptr(), q_rope.data_ptr(),
                                     stream_ptr(q.device)), "fl_mla_quant_q_store_k")
    return q_nope, q_scale, q_rope


class _FlMlaAbsorbArgs(ctypes.Structure):
    _fields_ = [("q", ctypes.c_void_p), ("q_stride_token", ctypes.c_int64), ("q_stride_head", ctypes.c_int64),
                ("num_tokens", ctypes.c_int64), ("num_heads", ctypes.c_int32), ("d_nope", ctypes.c_int32), ("d_rope", ctypes.c_int32),
                ("d_lora", ctypes.c_int32), ("w_kc", ctypes.c_void_p), ("w_stride_head", ctypes.c_int64),
                ("positions", ctypes.c_void_p), ("cos_sin_cache", ctypes.c_void_p), ("max_position", ctypes.c_int64),
                ("is_neox", ctypes.c_int32), ("latent", ctypes.c_void_p), ("latent_stride", ctypes.c_int64),
                ("cache_loc", ctypes.c_void_p), ("k_lora_cache", ctypes.c_void_p), ("k_scale_cache", ctypes.c_void_p),
                ("k_rope_cache", ctypes.c_void_p), ("num_slots", ctypes.c_int64), ("q_nope_out", ctypes.c_void_p),
                ("q_scale_out", ctypes.c_void_p), ("q_rope_out", ctypes.c_void_p)]


lib.fl_mla_absorb_rope_quant.argtypes = [ctypes.POINTER(_FlMlaAbsorbArgs), ctypes.c_void_p]
lib.fl_mla_absorb_rope_quant.restype = ctypes.c_int


def absorb_rope_quant(q: torch.Tensor, w_kc: torch.Tensor, positions: torch.Tensor, cos_sin_cache: torch.Tensor,
                      latent_cache: torch.Tensor = None, k_lora_cache: torch.Tensor = None, k_lora_scale_cache: torch.Tensor = None,
                      k_rope_cache: torch.Tensor = None, indices: torch.Tensor = None, is_neox: bool = False):
    """The query side of DeepseekV2AttentionMLA.forward_absorb_prepare + FlashMLABackend.forward_decode's two quantisers in ONE launch
    (srt/models/deepseek_v2.py:830-861, flashmla_backend.py:188-206): `torch.bmm(q_nope.transpose(0,1), w_kc)`, the rotary embedding of
    q_pe and k_pe, `quantize_and_cache_k` and `quantize_ckv_per_token_head` — the bf16 absorbed query is never written.
      q [T, H, 192] bf16 (nope 128 | rope 64; any token / head strides), w_kc [H, 128, 512] as the model holds it (the transposed view
      of a k-contiguous [H, 512, 128], deepseek_v2.py:1632), positions [T], cos_sin_cache [max_position, 64] f32,
      latent_cache [T, 576] bf16 (k_nope | k_pe — k_pe is rotated IN PLACE like the reference's call) with the three cache tensors and
      `indices` (cache locations): optional as a group.
    Returns (q_nope fp8 [T, H, 512], q_scale f32 [T, H, 1], q_rope bf16 [T, H, 64]) — the inputs of flash_mla_ckv_fp8_per_token —
    bit-identical to the four-launch chain.  An extension over the reference's module."""
    _req(q.is_cuda and q.dtype == torch.bfloat16 and q.dim() == 3 and q.shape[2] == 192 and q.stride(2) == 1, "q must be bf16 [T, H, 192]")
    T, H = q.shape[0], q.shape[1]
    _req(w_kc.is_cuda and w_kc.dtype == torch.bfloat16 and tuple(w_kc.shape) == (H, 128, 512) and w_kc.stride(1) == 1 and w_kc.stride(2) == 128,
         "w_kc must be the [H, 128, 512] view of a k-contiguous [H, 512, 128] bf16 tensor")
    _req(cos_sin_cache.is_cuda and cos_sin_cache.dtype == torch.float32 and cos_sin_cache.dim() == 2 and cos_sin_cache.shape[1] == 64
         and cos_sin_cache.is_contiguous(), "cos_sin_cache must be contiguous f32 [max_position, 64]")
    pos = positions.reshape(-1).to(torch.int64).contiguous()
    _req(pos.numel() == T and pos.is_cuda, "positions must hold one entry per token")
    a = _FlMlaAbsorbArgs()
    a.q, a.q_stride_token, a.q_stride_head, a.num_tokens, a.num_heads = q.data_ptr(), q.stride(0), q.stride(1), T, H
    a.d_nope, a.d_rope, a.d_lora = 128, 64, 512
    a.w_kc, a.w_stride_head = w_kc.data_ptr(), w_kc.stride(0)
    a.positions, a.cos_sin_cache, a.max_position, a.is_neox = pos.data_ptr(), cos_sin_cache.data_ptr(), cos_sin_cache.shape[0], int(bool(is_neox))
    if latent_cache is not None:
        for t, n in ((k_lora_cache, "k_lora_cache"), (k_lora_scale_cache, "k_lora_scale_cache"), (k_rope_cache, "k_rope_cache"), (indices, "indices")):
            _req(t is not None, f"{n} is required with latent_cache")
            _cuda_contig(t, n)
        lat = latent_cache.view(-1, latent_cache.shape[-1])
        _req(lat.is_cuda and lat.dtype == torch.bfloat16 and lat.shape == (T, 576) and lat.stride(1) == 1, "latent_cache must be bf16 [T, 576]")
        _req(k_lora_cache.dtype in _ONE_BYTE and k_lora_scale_cache.dtype == torch.float32 and k_rope_cache.dtype == torch.bfloat16,
             "bad cache dtypes")
        _req(indices.dtype == torch.int32 and indices.numel() == T, "indices must be int32 [T]")
        a.latent, a.latent_stride, a.cache_loc = lat.data_ptr(), lat.stride(0), indices.data_ptr()
        a.k_lora_cache, a.k_scale_cache, a.k_rope_cache = k_lora_cache.data_ptr(), k_lora_scale_cache.data_ptr(), k_rope_cache.data_ptr()
        a.num_slots = k_lora_cache.numel() // 512
    q_nope = torch.empty(T, H, 512, dtype=torch.float8_e4m3fn, device=q.device)
    q_scale = torch.empty(T, H, 1, dtype=torch.float32, device=q.device)
    q_rope = torch.empty(T, H, 64, dtype=torch.bfloat16, device=q.device)
    a.q_nope_out, a.q_scale_out, a.q_rope_out = q_nope.data_ptr(), q_scale.data_ptr(), q_rope.data_ptr()
    check(lib.fl_mla_absorb_rope_quant(ctypes.byref(a), stream_ptr(q.device)), "fl_mla_absorb_rope_quant")
    return q_nope, q_scale, q_rope


def dequantize_ckv_fused_indexed(k_lora_fp8: torch.Tensor, k_rope: torch.Tensor, k_scale: torch.Tensor,
                                 indices: torch.Tensor):
    """-> (k_lora_deq bf16 [n,1,512], k_rope_deq bf16 [n,1,64]) gathered at `indices`."""
    for t, n in ((k_lora_fp8, "k_lora_fp8"), (k_rope, "k_rope"), (k_scale, "k_scale"), (indices, "indices")):
        _cuda_contig(t, n)
    _req(k_lora_fp8.dtype in _ONE_BYTE and k_rope.dtype == torch.bfloat16 and k_scale.dtype == torch.float32,
         "bad cache dtypes")
    idx = indices if indices.dtype == torch.int32 else indices.to(torch.int32)
    n = idx.numel()
    d_nope, d_rope = k_lora_fp8.shape[-1], k_rope.shape[-1]
    num_slots = k_lora_fp8.numel() // d_nope
    lora = torch.empty((n,) + tuple(k_lora_fp8.shape[1:]), dtype=torch.bfloat16, device=k_rope.device)
    rope = torch.empty((n,) + tuple(k_rope.shape[1:]), dtype=torch.bfloat16, device=k_rope.device)
    check(lib.fl_mla_dequant_gather(k_lora_fp8.data_ptr(), k_rope.data_ptr(), k_scale.data_ptr(), idx.data_ptr(), n,
                                    d_nope, d_rope, num_slots, lora.data_ptr(), rope.data_ptr(),
                                    stream_ptr(k_rope.device)), "fl_mla_dequant_gather")
    return lora, rope


lib.fl_mla_set_merge_timeout.argtypes = [ctypes.c_double]
lib.fl_mla_set_merge_timeout.restype = ctypes.c_int
lib.fl_mla_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
lib.fl_mla_workspace_bytes.restype = ctypes.c_int


def _decode(args: FlMlaDecodeArgs, dev):
    check(lib.fl_mla_decode(ctypes.byref(args), stream_ptr(dev)), "fl_mla_decode")


def _common(args, q_like, block_table, cache_seqlens, tile_scheduler_metadata, num_splits, softmax_scale, causal):
    bs, s_q, h_q = q_like.shape[0], q_like.shape[1], q_like.shape[2]
    dev = q_like.device
    for t, n in ((block_table, "block_table"), (cache_seqlens, "cache_seqlens"),
                 (tile_scheduler_metadata, "tile_scheduler_metadata"), (num_splits, "num_splits")):
        _req(t.is_cuda and t.dtype == torch.int32, f"{n} must be an int32 device tensor")
    _req(block_table.dim() == 2 and block_table.stride(1) == 1, "block_table must be [bs, pages] with unit inner stride")
    _req(cache_seqlens.is_contiguous() and tile_scheduler_metadata.is_contiguous() and num_splits.is_contiguous(),
         "metadata tensors must be contiguous")
    _req(block_table.shape[0] >= bs and cache_seqlens.shape[0] >= bs and num_splits.shape[0] >= bs + 1,
         "metadata tensors shorter than batch")
    num_parts = tile_scheduler_metadata.shape[0]
    rows = s_q * h_q
    args.bs, args.s_q, args.h_q = bs, s_q, h_q
    args.causal = 1 if causal else 0
    args.num_parts = num_parts
    args.softmax_scale = float(softmax_scale)
    args.block_table = block_table.data_ptr()
    args.block_table_stride = block_table.stride(0)
    args.block_table_cols = block_table.shape[1]
    args.cache_seqlens = cache_seqlens.data_ptr()
    args.tile_scheduler_metadata = tile_scheduler_metadata.data_ptr()
    args.num_splits = num_splits.data_ptr()
    out = torch.empty((bs, s_q, h_q, 512), dtype=torch.bfloat16, device=dev)
    lse = torch.empty((bs, h_q, s_q), dtype=torch.float32, device=dev)
    ob, lb = ctypes.c_int64(), ctypes.c_int64()
    check(lib.fl_mla_workspace_bytes(int(args.kv_format), bs, s_q, h_q, num_parts, ctypes.byref(ob), ctypes.byref(lb)),
          "fl_mla_workspace_bytes")
    o_accum = torch.empty((ob.value,), dtype=torch.uint8, device=dev)       # bf16 or f32 partial rows: the library says which
    lse_accum = torch.empty((lb.value,), dtype=torch.uint8, device=dev)
    args.out, args.lse, args.o_accum, args.lse_accum = out.data_ptr(), lse.data_ptr(), o_accum.data_ptr(), lse_accum.data_ptr()
    return out, lse, (o_accum, lse_accum)


def flash_mla_ckv_fp8_per_token(q_nope: torch.Tensor, q_rope: torch.Tensor, k_cache_lora: torch.Tensor,
                                k_cache_rope: torch.Tensor, q_scale: torch.Tensor, k_scale: torch.Tensor,
                                block_table: torch.Tensor, cache_seqlens: torch.Tensor, head_dim_v: int,
                                tile_scheduler_metadata: torch.Tensor, num_splits: torch.Tensor,
                                softmax_scale: Optional[float] = None, causal: bool = False):
    """Paged MLA decode over the per-token-FP8 latent cache -> (o bf16 [bs,s_q,H,512], lse f32 [bs,H,s_q]).
    q_nope fp8 [bs,s_q,H,512], q_scale f32 [bs,s_q,H,1], q_rope bf16 [bs,s_q,H,64];
    k_cache_lora u8/fp8 [pages,64,1,512], k_scale f32 [pages,64,1,1], k_cache_rope bf16 [pages,64,1,64]."""
    for t, n in ((q_nope, "q_nope"), (q_rope, "q_rope"), (q_scale, "q_scale"), (k_cache_lora, "k_cache_lora"),
                 (k_cache_rope, "k_cache_rope"), (k_scale, "k_scale")):
        _cuda_contig(t, n)
    _req(q_nope.dim() == 4 and q_nope.dtype in _ONE_BYTE, "q_nope must be fp8 [bs,s_q,H,512]")
    _req(q_rope.dtype == torch.bfloat16 and k_cache_rope.dtype == torch.bfloat16, "rope tensors must be bfloat16")
    _req(q_scale.dtype == torch.float32 and k_scale.dtype == torch.float32, "scales must be float32")
    _req(k_cache_lora.dtype in _ONE_BYTE, "k_cache_lora must be a 1-byte dtype")
    _req(head_dim_v == 512 and q_nope.shape[-1] == 512 and q_rope.shape[-1] == 64, "only kv_lora=512, rope=64")
    _req(k_cache_lora.shape[1] == PAGE_SIZE, "page size must be 64")
    if softmax_scale is None:
        softmax_scale = (q_nope.shape[-1] + q_rope.shape[-1]) ** -0.5
    a = FlMlaDecodeArgs()
    a.struct_bytes = ctypes.sizeof(FlMlaDecodeArgs)
    a.kv_format = KV_FP8_PER_TOKEN
    a.d_nope, a.d_rope = 512, 64
    a.q_nope, a.q_rope, a.q_scale = q_nope.data_ptr(), q_rope.data_ptr(), q_scale.data_ptr()
    a.k_nope, a.k_rope, a.k_scale = k_cache_lora.data_ptr(), k_cache_rope.data_ptr(), k_scale.data_ptr()
    a.num_pages = k_cache_lora.shape[0]
    out, lse, _ws = _common(a, q_nope, block_table, cache_seqlens, tile_scheduler_metadata, num_splits, softmax_scale, causal)
    _decode(a, q_nope.device)
    return out, lse


def flash_mla_ckv_fp8_per_token_bf16_q(q: torch.Tensor, k_cache_lora: torch.Tensor, k_cache_rope: torch.Tensor, k_scale: torch.Tensor,
                                       block_table: torch.Tensor, cache_seqlens: torch.Tensor, head_dim_v: int,
                                       tile_scheduler_metadata: torch.Tensor, num_splits: torch.Tensor,
                                       softmax_scale: Optional[float] = None, causal: bool = False):
    """`flash_mla_ckv_fp8_per_token(*quantize_ckv_per_token_head(q, 512), ...)` in ONE launch: the decode kernel quantises the query
    (K4, flashmla_backend.py:198-206) in its request prologue — bit-identical output, one launch and a write + re-read of the quantised
    query less.  q bf16 [bs, s_q, H, 576] (the absorbed query as forward_absorb_prepare builds it), more than 32 query rows per request.
    An extension over the reference's module."""
    for t, n in ((q, "q"), (k_cache_lora, "k_cache_lora"), (k_cache_rope, "k_cache_rope"), (k_scale, "k_scale")):
        _cuda_contig(t, n)
    _req(q.dim() == 4 and q.dtype == torch.bfloat16 and q.shape[-1] == 576 and head_dim_v == 512, "q must be bf16 [bs,s_q,H,576]")
    _req(q.shape[1] * q.shape[2] > 32, "the fused-K4 decode serves more than 32 query rows per request (use the two-call form below that)")
    _req(k_cache_rope.dtype == torch.bfloat16 and k_scale.dtype == torch.float32 and k_cache_lora.dtype in _ONE_BYTE, "bad cache dtypes")
    _req(k_cache_lora.shape[1] == PAGE_SIZE, "page size must be 64")
    if softmax_scale is None:
        softmax_scale = 576 ** -0.5
    a = FlMlaDecodeArgs()
    a.struct_bytes = ctypes.sizeof(FlMlaDecodeArgs)
    a.kv_format = KV_FP8_PER_TOKEN
    a.d_nope, a.d_rope = 512, 64
    a.q_bf16 = q.data_ptr()
    a.k_nope, a.k_rope, a.k_scale = k_cache_lora.data_ptr(), k_cache_rope.data_ptr(), k_scale.data_ptr()
    a.num_pages = k_cache_lora.shape[0]
    out, lse, _ws = _common(a, q, block_table, cache_seqlens, tile_scheduler_metadata, num_splits, softmax_scale, causal)
    _decode(a, q.device)
    return out, lse


def flash_mla_with_kvcache(q: torch.Tensor, k_cache: torch.Tensor, block_table: torch.Tensor,
                           cache_seqlens: torch.Tensor, head_dim_v: int, tile_scheduler_metadata: torch.Tensor,
                           num_splits: torch.Tensor, softmax_scale: Optional[float] = None, causal: bool = False,
                           descale_q: Optional[torch.Tensor] = None, descale_k: Optional[torch.Tensor] = None):
    """Paged MLA decode over a single 576-wide cache tensor (bf16, or plain fp8 with scalar descales)
    (flashmla_backend.py:145-175,227-254) -> (o, lse)."""
    _cuda_contig(q, "q")
    _cuda_contig(k_cache, "k_cache")
    _req(q.dim() == 4 and q.shape[-1] == 576 and head_dim_v == 512, "only head_dim 576 / head_dim_v 512")
    _req(k_cache.shape[1] == PAGE_SIZE, "page size must be 64")
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** -0.5
    a = FlMlaDecodeArgs()
    a.struct_bytes = ctypes.sizeof(FlMlaDecodeArgs)
    a.d_nope, a.d_rope = 512, 64
    if q.dtype == torch.bfloat16:
        _req(k_cache.dtype == torch.bfloat16, "bf16 q needs a bf16 cache")
        a.kv_format = KV_BF16_576
    else:
        _req(q.dtype in _ONE_BYTE and k_cache.dtype in _ONE_BYTE, "fp8 q needs an fp8 cache")
        a.kv_format = KV_FP8_576
        # the reference passes device tensors torch.ones(1) (flashmla_backend.py:237-238): forwarded as device pointers,
        # read inside the kernel (no .item(): the call stays graph-capturable)
        for t in (descale_q, descale_k):
            _req(t is None or (t.is_cuda and t.dtype == torch.float32 and t.numel() >= 1), "descales must be f32 device tensors")
        a.descale_q = descale_q.data_ptr() if descale_q is not None else None
        a.descale_k = descale_k.data_ptr() if descale_k is not None else None
    a.q_nope = q.data_ptr()
    a.k_nope = k_cache.data_ptr()
    a.num_pages = k_cache.shape[0]
    out, lse, _ws = _common(a, q, block_table, cache_seqlens, tile_scheduler_metadata, num_splits, softmax_scale, causal)
    _decode(a, q.device)
    return out, lse


def set_merge_timeout(seconds: float) -> None:
    """Budget of the in-kernel split merge's wait for a request's other pieces (default 2 s; env FLUENT_MLA_MERGE_TIMEOUT_S).  A merger that
    gives up poisons its rows with NaN and the NEXT decode call raises (include/fluent_mi355.h: fl_mla_set_merge_timeout).  Synchronises."""
    check(lib.fl_mla_set_merge_timeout(float(seconds)), "fl_mla_set_merge_timeout")
