"""`flashinfer.comm.vllm_ar` (C4: the plain custom all-reduce of python/sglang/srt/_custom_ops.py:11-60, driven by
distributed/device_communicators/custom_all_reduce.py:42-360).

The reference's implementation registers cudaIpc-mapped buffers of all peers and reduces them in one kernel.  Its Python
driver `CustomAllreduce` needs `CudaRTLibrary` (libcudart through ctypes, custom_all_reduce.py:18,212-229) for the handle
exchange — absent on ROCm — and is disabled there (`--disable-custom-all-reduce`; `GroupCoordinator.all_reduce` then uses
pynccl = RCCL, parallel_state.py).  This module keeps the ENTRY POINTS with the reference's signatures; behind them:
  * decode-sized bf16 inputs ([rows <= 256, H <= 8192], the shape of the hidden states the
    reference's custom all-reduce serves) run as ONE peer-mapped one-shot kernel (csrc/comm_oneshot.hip: the same transport
    as the fused C5 — fl_allreduce_fused with no residual and no norm is a plain one-shot sum; the workspace and the hipIpc
    mappings are set up by `init_custom_ar`, a collective call like the reference's);
  * everything else (other dtypes / sizes, no HIP device, gloo groups) sums through `torch.distributed` on the current
    stream (RCCL; graph-capturable)."""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist

_handles = {}
_next = [1]
_default_group = [None]


def set_group(group) -> None:
    """The process group custom all-reduces run on (default: the world group)."""
    _default_group[0] = group


def meta_size() -> int:
    return 0   # no signal / metadata area: nothing is IPC-mapped on this route


def _make_oneshot(group):
    """the one-shot communicator behind a custom-all-reduce handle (None: RCCL only).  Collective over `group`."""
    want = os.environ.get("FLUENT_ONESHOT", "auto")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    multi = dist.is_initialized() and world > 1 and dist.get_backend(group) == "nccl"
    if want == "0" or not torch.cuda.is_available() or not (multi or want == "1"):
        return None
    from fluent_mi355.oneshot import OneShotComm

    try:
        return OneShotComm(dist.get_rank(group) if multi else 0, world if multi else 1,
                           256, 8192, group=group)
    except RuntimeError as ex:
        import warnings

        warnings.warn(f"flashinfer.comm.vllm_ar: one-shot all-reduce unavailable ({ex}); using RCCL")
        return None


def init_custom_ar(ipc_tensors: List[int], rank_data: torch.Tensor, rank: int, full_nvlink: bool) -> int:
    h = _next[0]
    _next[0] += 1
    _handles[h] = {"rank": rank, "group": _default_group[0], "buffers": list(ipc_tensors),
                   "oneshot": _make_oneshot(_default_group[0])}
    return h


def dispose(fa: int) -> None:
    h = _handles.pop(fa, None)
    if h is not None and h.get("oneshot") is not None:
        h["oneshot"].close()


def register_buffer(fa: int, ipc_tensors: List[int]) -> None:
    _handles[fa]["buffers"] = list(ipc_tensors)


def get_graph_buffer_ipc_meta(fa: int) -> Tuple[List[int], List[int]]:
    return [], []


def register_graph_buffers(fa: int, handles: List[List[int]], offsets: List[List[int]]) -> None:
    return None


def all_reduce(fa: int, inp: torch.Tensor, out: torch.Tensor, reg_buffer: int = 0, reg_buffer_sz_bytes: int = 0,
               num_ctas: int = 4) -> None:
    """out = sum over the group of inp (custom_all_reduce.py:319-341)."""
    if fa not in _handles:
        raise RuntimeError("vllm_ar.all_reduce: unknown handle (init_custom_ar was not called)")
    osc = _handles[fa].get("oneshot")
    if osc is not None and inp.dim() >= 1 and inp.numel() > 0 and out.shape == inp.shape:
        H = inp.shape[-1]
        x, o = inp.reshape(-1, H), out.reshape(-1, H)
        if (x.data_ptr() == inp.data_ptr() and o.data_ptr() == out.data_ptr() and osc.fits(x.shape[0], H)
                and osc.accepts(x, False, residual_out=o)):
            osc.allreduce_fused(x, residual_out=o)   # no residual, no norm: the one-shot SUM (in place is fine: rows are
            return                                   # pushed to the peers' inboxes before anything is written)
    if out.data_ptr() != inp.data_ptr():
        out.copy_(inp)
    if dist.is_initialized() and dist.get_world_size(_handles[fa]["group"]) > 1:
        dist.all_reduce(out, group=_handles[fa]["group"])
