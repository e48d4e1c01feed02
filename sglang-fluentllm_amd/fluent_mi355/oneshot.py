"""One-shot peer-mapped fused collectives (C5/C6) — host side of csrc/comm_oneshot.hip.

Replaces the IPC workspace + one-shot kernels of the reference's flashinfer.comm
(srt/layers/flashinfer_comm_fusion.py:64-109 workspace creation, :286-401 all-reduce fusion, :404-513 reduce-scatter
fusion): every rank allocates one uncached workspace, the 64-byte hipIpc handles are exchanged once through
torch.distributed (`all_gather_object`: setup only, never on the hot path), every rank maps its peers' workspaces, and
from then on an all-reduce / reduce-scatter + residual + RMSNorm (+ 1x128 fp8 quant) is ONE kernel launch with no RCCL
call.  `fluent_mi355.comm` routes to it for token counts up to `max_tokens` (env FLUENT_ONESHOT=0 disables)."""
from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist

from ._lib import check, lib, stream_ptr

MAX_ONESHOT_TOKENS = 1024

_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
lib.fl_comm_create.argtypes = [_i32, _i32, _i64, _i32, ctypes.POINTER(_vp)]
lib.fl_comm_local_handle.argtypes = [_vp, _vp]
lib.fl_comm_connect.argtypes = [_vp, _vp]
lib.fl_comm_set_timeout.argtypes = [_vp, ctypes.c_double]
lib.fl_allreduce_fused.argtypes = [_vp, _vp, _i64, _i32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _i64, _i64, _vp]
lib.fl_reducescatter_fused.argtypes = [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _i64, _i64, _vp]
lib.fl_allgather_fused.argtypes = [_vp, _vp, _i64, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _i64, _i64, _vp]
lib.fl_alltoall_oneshot.argtypes = [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]
lib.fl_alltoall_oneshot.restype = _i32
lib.fl_comm_check.argtypes = [_vp]
lib.fl_comm_destroy.argtypes = [_vp]
for _n in ("fl_comm_create", "fl_comm_local_handle", "fl_comm_connect", "fl_comm_set_timeout", "fl_allreduce_fused",
           "fl_reducescatter_fused", "fl_allgather_fused", "fl_comm_check", "fl_comm_destroy"):
    getattr(lib, _n).restype = _i32


def _p(t):
    return None if t is None else t.data_ptr()


class OneShotComm:
    """One rank's end of the peer-mapped exchange.  `exchange` maps this rank's 64-byte handle to the list of all ranks'
    handles in rank order (default: all_gather_object on `group`)."""

    def __init__(self, rank, world, max_tokens, hidden, group=None, exchange=None, timeout_s=None):
        """Collective over the group: every rank runs the same two exchanges whether or not its local steps succeed, and
        either all ranks end up connected or ALL raise RuntimeError (nobody is left on a different route)."""
        self.rank, self.world, self.hidden = int(rank), int(world), int(hidden)
        self.max_tokens = min(int(max_tokens), MAX_ONESHOT_TOKENS)
        if timeout_s is None and os.environ.get("FLUENT_ONESHOT_TIMEOUT_S"):
            timeout_s = float(os.environ["FLUENT_ONESHOT_TIMEOUT_S"])   # default (C side): 10 s
        self._h = None
        if exchange is None:
            def exchange(obj):
                out = [None] * self.world
                dist.all_gather_object(out, obj, group=group)
                return out
        err = None
        mine = ctypes.create_string_buffer(64)
        try:
            h = _vp()
            check(lib.fl_comm_create(self.rank, self.world, self.max_tokens, self.hidden, ctypes.byref(h)), "fl_comm_create")
            self._h = h
            if self.world > 1:
                check(lib.fl_comm_local_handle(self._h, ctypes.cast(mine, _vp)), "fl_comm_local_handle")
        except RuntimeError as ex:
            err = str(ex)
        if self.world > 1:
            got = exchange((bytes(mine.raw), err))                       # exchange 1: handles (+ who failed so far)
            bad = [f"rank {r}: {e}" for r, (_, e) in enumerate(got) if e is not None]
            if not bad:
                try:
                    blob = ctypes.create_string_buffer(b"".join(hd for hd, _ in got), 64 * self.world)
                    check(lib.fl_comm_connect(self._h, ctypes.cast(blob, _vp)), "fl_comm_connect")
                except RuntimeError as ex:
                    err = str(ex)
                bad = [f"rank {r}: {e}" for r, e in enumerate(exchange(err)) if e is not None]   # exchange 2: who connected
            if bad:
                self.close()
                raise RuntimeError("one-shot comm setup failed on " + "; ".join(bad))
        elif err is not None:
            raise RuntimeError(err)
        if timeout_s is not None:
            check(lib.fl_comm_set_timeout(self._h, float(timeout_s)), "fl_comm_set_timeout")

    def fits(self, tokens, hidden, reduce_scatter=False):
        per = -(-tokens // self.world) if reduce_scatter else tokens
        return tokens <= MAX_ONESHOT_TOKENS and per <= self.max_tokens and hidden <= self.hidden and hidden % 8 == 0

    @staticmethod
    def _strides(scale_out):
        return (0, 0) if scale_out is None else (scale_out.stride(0), scale_out.stride(1))

    def slice_rows(self, tokens, reduce_scatter):
        """rows of the per-rank tensors of an operation over `tokens` rows (get_num_tokens_per_rank for a reduce-scatter)"""
        if not reduce_scatter:
            return tokens
        return tokens // self.world + (1 if self.rank < tokens % self.world else 0)

    def accepts(self, x, reduce_scatter=False, add_in=None, residual_in=None, gamma=None, residual_out=None, norm_out=None,
                quant_out=None, scale_out=None):
        """What the kernel assumes about the tensors it is handed raw pointers of (the same contract HipNormOps.add_rmsnorm
        asserts on the RCCL route): bf16, contiguous, on this device, [rows, H] with rows = this rank's slice for the
        per-rank tensors; quant_out 1-byte [rows, H]; scale_out f32 2-D covering [rows, H/128].  False -> take the RCCL
        route (comm.py) instead of misreading memory."""
        if x.dim() != 2 or x.dtype != torch.bfloat16 or not x.is_cuda or not x.is_contiguous():
            return False
        T, H = x.shape
        rows = self.slice_rows(T, reduce_scatter)
        for t in (add_in, residual_in, residual_out, norm_out):
            if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or t.device != x.device
                                  or t.dim() != 2 or t.shape[0] < rows or t.shape[1] != H):
                return False
        if gamma is not None and (gamma.dtype != torch.bfloat16 or not gamma.is_contiguous() or gamma.device != x.device
                                  or gamma.numel() != H):
            return False
        if quant_out is not None:
            if (quant_out.element_size() != 1 or not quant_out.is_contiguous() or quant_out.device != x.device
                    or quant_out.dim() != 2 or quant_out.shape[0] < rows or quant_out.shape[1] != H or H % 128):
                return False
            if (scale_out is None or scale_out.dtype != torch.float32 or scale_out.device != x.device or scale_out.dim() != 2
                    or scale_out.shape[0] < rows or scale_out.shape[1] < H // 128):
                return False
        return True

    def _require(self, ok, what):
        if not ok:
            raise ValueError(f"one-shot comm {what}: tensors must be bf16, contiguous, on one device, [rows, H] with this "
                             "rank's row count (fp8 quant_out [rows, H], f32 scale_out [rows, >= H/128]); use the RCCL route")

    def allreduce_fused(self, x, residual_in=None, gamma=None, eps=1e-6, residual_out=None, norm_out=None, quant_out=None,
                        scale_out=None):
        T, H = x.shape
        if self.fits(T, H):   # (beyond the workspace the C side reports "exceeds the workspace")
            self._require(self.accepts(x, False, None, residual_in, gamma, residual_out, norm_out, quant_out, scale_out), "allreduce_fused")
        st, sg = self._strides(scale_out)
        check(lib.fl_allreduce_fused(self._h, x.data_ptr(), T, H, _p(residual_in), _p(gamma), float(eps), _p(residual_out),
                                     _p(norm_out), _p(quant_out), _p(scale_out), st, sg, stream_ptr(x.device)), "fl_allreduce_fused")

    def reducescatter_fused(self, x, add_in=None, residual_in=None, gamma=None, eps=1e-6, residual_out=None, norm_out=None,
                            quant_out=None, scale_out=None):
        T, H = x.shape
        if self.fits(T, H, reduce_scatter=True):
            self._require(self.accepts(x, True, add_in, residual_in, gamma, residual_out, norm_out, quant_out, scale_out),
                          "reducescatter_fused")
        st, sg = self._strides(scale_out)
        check(lib.fl_reducescatter_fused(self._h, x.data_ptr(), T, H, _p(add_in), _p(residual_in), _p(gamma), float(eps),
                                         _p(residual_out), _p(norm_out), _p(quant_out), _p(scale_out), st, sg, stream_ptr(x.device)),
              "fl_reducescatter_fused")

    def accepts_gather(self, x, total, out, x_norm_out=None, quant_out=None, scale_out=None, q_rank=0, kv_rank=0,
                       gamma_q=None, gamma_kv=None):
        """the raw-pointer contract of allgather_fused: bf16 contiguous [rows of this rank, D] -> out [total, D] (bf16,
        contiguous, same device); dual-norm outputs [total, q_rank]"""
        if x.dim() != 2 or x.dtype != torch.bfloat16 or not x.is_cuda or not x.is_contiguous():
            return False
        D = x.shape[1]
        if x.shape[0] != self.slice_rows(total, True) or total > MAX_ONESHOT_TOKENS or -(-total // self.world) > self.max_tokens:
            return False
        if D % 8 or D > self.hidden:
            return False
        if (out.dtype != torch.bfloat16 or not out.is_contiguous() or out.device != x.device or out.dim() != 2
                or out.shape[0] < total or out.shape[1] != D):
            return False
        if q_rank:
            for g_, n_ in ((gamma_q, q_rank), (gamma_kv, kv_rank)):
                if g_ is None or g_.dtype != torch.bfloat16 or not g_.is_contiguous() or g_.device != x.device or g_.numel() != n_:
                    return False
            if x_norm_out is not None and (x_norm_out.dtype != torch.bfloat16 or not x_norm_out.is_contiguous()
                                           or x_norm_out.device != x.device or x_norm_out.dim() != 2
                                           or x_norm_out.shape[0] < total or x_norm_out.shape[1] != q_rank):
                return False
            if quant_out is not None:
                if (quant_out.element_size() != 1 or not quant_out.is_contiguous() or quant_out.device != x.device
                        or quant_out.dim() != 2 or quant_out.shape[0] < total or quant_out.shape[1] != q_rank or q_rank % 128):
                    return False
                if (scale_out is None or scale_out.dtype != torch.float32 or scale_out.device != x.device or scale_out.dim() != 2
                        or scale_out.shape[0] < total or scale_out.shape[1] < q_rank // 128):
                    return False
        return True

    def allgather_fused(self, x, total, out, q_rank=0, kv_rank=0, gamma_q=None, gamma_kv=None, eps_q=1e-6, eps_kv=1e-6,
                        x_norm_out=None, quant_out=None, scale_out=None):
        """every rank's rows `x` [get_num_tokens_per_rank(world, total)[rank], D] -> out[:total] in rank order, one launch;
        q_rank > 0: + the dual RMSNorm of C7 on every gathered row (csrc/comm_oneshot.hip oneshot_ag_kernel)"""
        self._require(self.accepts_gather(x, total, out, x_norm_out, quant_out, scale_out, q_rank, kv_rank, gamma_q, gamma_kv),
                      "allgather_fused")
        st, sg = self._strides(scale_out)
        check(lib.fl_allgather_fused(self._h, x.data_ptr() if x.numel() else None, x.shape[0], int(total), x.shape[1], out.data_ptr(),
                                     int(q_rank), int(kv_rank), _p(gamma_q), _p(gamma_kv), float(eps_q), float(eps_kv),
                                     _p(x_norm_out), _p(quant_out), _p(scale_out), st, sg, stream_ptr(x.device)), "fl_allgather_fused")

    def accepts_alltoall(self, send, recv, cap):
        """raw-pointer contract of alltoall: two distinct contiguous device tensors [world * cap, X] of one dtype whose rows are a
        multiple of 16 bytes and fit the workspace rows; cap rows per peer fit the inbox and one launch"""
        if send.dim() != 2 or not send.is_cuda or not send.is_contiguous() or recv.shape != send.shape or recv.dtype != send.dtype \
                or recv.device != send.device or not recv.is_contiguous() or recv.data_ptr() == send.data_ptr():
            return False
        row_bytes = send.shape[1] * send.element_size()
        return (send.shape[0] == self.world * cap and 1 <= cap <= self.max_tokens and cap * self.world <= MAX_ONESHOT_TOKENS
                and row_bytes % 16 == 0 and row_bytes <= 2 * self.hidden)

    def alltoall(self, send, recv, cap, ids_col=-1, top_k=0):
        """equal-split all-to-all of `world` slabs of `cap` rows (slab p of `send` -> slab `rank` of rank p's `recv`), one launch.
        ids_col: index (in 2-byte elements) of a row's top_k int32 expert ids, or -1; rows with no id >= 0 move their tail only."""
        self._require(self.accepts_alltoall(send, recv, cap), "alltoall")
        D = send.shape[1] * send.element_size() // 2
        check(lib.fl_alltoall_oneshot(self._h, send.data_ptr(), recv.data_ptr(), int(cap), int(D), int(ids_col), int(top_k),
                                      stream_ptr(send.device)), "fl_alltoall_oneshot")

    def check(self):
        """synchronises; raises if a flag wait ever timed out (a peer died or issued a different sequence of operations)"""
        check(lib.fl_comm_check(self._h), "fl_comm_check")

    def close(self):
        if self._h is not None:
            lib.fl_comm_destroy(self._h)
            self._h = None
