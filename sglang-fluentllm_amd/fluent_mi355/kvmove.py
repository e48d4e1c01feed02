"""Host-side mirror of the KV row move of the reference's pools: `copy_all_layer_kv_cache_tiled[grid](data_ptrs,
data_strides, tgt_loc, src_loc, ...)` as `MLATokenToKVPool.move_kv_cache` launches it
(python/sglang/srt/mem_cache/memory_pool.py:746-777, kernel :2055-2090), one HIP launch for all layers' buffers
(csrc/kv_move.hip).  No fallback."""
from __future__ import annotations

import ctypes

import torch

from ._lib import check, lib, stream_ptr

_i64, _vp, _i = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
lib.fl_kv_move.argtypes = [_vp, _vp, _i, _i64, _vp, _vp, _i64, _i64, _vp]
lib.fl_kv_move.restype = _i
lib.fl_kv_move_staged.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp]
lib.fl_kv_move_staged.restype = _i
_MAX_ROWS_IN_REGISTERS = 8192   # csrc/kv_move.hip: kCap
_STAGING_MAX_BYTES = 4 << 30   # staged path (> 8192 rows): upper bound of the gather area of ONE move (61 layers x 656 B x 100 k rows = 4 GB)


class KVMoveTable:
    """Device table of (base pointer, bytes per row) over every buffer whose rows are indexed by the same slot ids — the
    reference builds the same two tensors once per pool (memory_pool.py:330-343: data_ptrs / data_strides)."""

    def __init__(self, buffers):
        buffers = list(buffers)
        if not buffers:
            raise RuntimeError("KVMoveTable: no buffers")
        dev = buffers[0].device
        rows = buffers[0].shape[0]
        for b in buffers:
            if not b.is_cuda or b.device != dev or not b.is_contiguous() or b.shape[0] != rows:
                raise RuntimeError("KVMoveTable: buffers must be contiguous CUDA/HIP tensors of one device with the same number of rows")
            if (b.stride(0) * b.element_size()) % 4:
                raise RuntimeError("KVMoveTable: rows must be multiples of 4 bytes")
        self.buffers = buffers   # keeps the storages alive
        self.num_slots = rows
        row_bytes = [b.stride(0) * b.element_size() for b in buffers]
        self.max_row_bytes = max(row_bytes)
        self.data_ptrs = torch.tensor([b.data_ptr() for b in buffers], dtype=torch.int64).to(dev)   # (addresses < 2^63)
        self.row_bytes = torch.tensor(row_bytes, dtype=torch.int64).to(dev)
        self.sum_row_bytes = sum(row_bytes)
        prefix = [0]
        for rb in row_bytes[:-1]:
            prefix.append(prefix[-1] + rb)
        self.row_prefix = torch.tensor(prefix, dtype=torch.int64).to(dev)   # staging layout of the > 8192-row path
        self._staging = {}   # stream -> staging area of the > 8192-row path

    def move(self, tgt_loc, src_loc):
        """rows tgt_loc[i] <- src_loc[i] in every buffer; sets may overlap (all reads precede all writes)"""
        if tgt_loc.numel() != src_loc.numel():
            raise RuntimeError("move_kv_cache: tgt_loc and src_loc differ in length")
        if tgt_loc.numel() == 0:
            return
        dev = self.data_ptrs.device
        t = tgt_loc.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        s = src_loc.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        n = t.numel()
        if n <= _MAX_ROWS_IN_REGISTERS:
            check(lib.fl_kv_move(self.data_ptrs.data_ptr(), self.row_bytes.data_ptr(), len(self.buffers), self.max_row_bytes,
                                 t.data_ptr(), s.data_ptr(), n, self.num_slots, stream_ptr(dev)), "fl_kv_move")
            return
        # more rows than one workgroup holds between its reads and its writes: gather every source row into a staging area, then scatter
        # (two launches, same semantics: ALL reads of the move precede ALL writes, so the area holds the whole move).  The area is bounded:
        # a move that would need more than _STAGING_MAX_BYTES is refused rather than silently pinning gigabytes.  Eager calls share
        # one area per stream, grown on demand and released when a much smaller move follows; a call made while the stream is CAPTURING
        # gets a fresh area from the graph's private pool (nothing allocated under one capture is reused by another: the bmm.py rule).
        need = n * self.sum_row_bytes
        if need > _STAGING_MAX_BYTES:
            raise RuntimeError(f"move_kv_cache: {n} rows x {self.sum_row_bytes} B = {need} B of staging exceed the {_STAGING_MAX_BYTES} B bound; "
                               "split the move into independent (non-overlapping) batches")
        if torch.cuda.is_current_stream_capturing():
            staging = torch.empty(need, dtype=torch.uint8, device=dev)
        else:
            key = stream_ptr(dev)
            staging = self._staging.get(key)
            if staging is None or staging.numel() < need or staging.numel() > 8 * max(need, 1 << 20):
                staging = self._staging[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        check(lib.fl_kv_move_staged(self.data_ptrs.data_ptr(), self.row_bytes.data_ptr(), self.row_prefix.data_ptr(), len(self.buffers),
                                    t.data_ptr(), s.data_ptr(), n, self.num_slots, staging.data_ptr(), staging.numel(),
                                    self.sum_row_bytes, stream_ptr(dev)), "fl_kv_move_staged")


def move_kv_cache(buffers, tgt_loc, src_loc):
    """one-shot form (builds the table per call; pools keep a KVMoveTable)"""
    KVMoveTable(buffers).move(tgt_loc, src_loc)
