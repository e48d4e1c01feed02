"""`flashinfer.comm.vllm_ar` (C4: the plain custom all-reduce of python/sglang/srt/_custom_ops.py:11-60, driven by
distributed/device_communicators/custom_all_reduce.py:42-360).

The reference's implementation registers cudaIpc-mapped buffers of all peers and reduces them in one kernel.  Its Python
driver `CustomAllreduce` needs `CudaRTLibrary` (libcudart through ctypes, custom_all_reduce.py:18,212-229) for the handle
exchange — absent on ROCm — and is disabled there (`--disable-custom-all-reduce`; `GroupCoordinator.all_reduce` then uses
pynccl = RCCL, parallel_state.py).  This module therefore keeps the ENTRY POINTS importable and functional over RCCL: a
handle is a small registry entry carrying the process group; `all_reduce` sums through `torch.distributed` on the current
stream (graph-capturable with RCCL).  The latency-critical fused variants (C5-C7) are `flashinfer.comm.trtllm_*_fusion`
(fluent_mi355/comm.py) and the peer-mapped one-shot kernels of fluent_mi355/oneshot.py."""
from __future__ import annotations

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


def init_custom_ar(ipc_tensors: List[int], rank_data: torch.Tensor, rank: int, full_nvlink: bool) -> int:
    h = _next[0]
    _next[0] += 1
    _handles[h] = {"rank": rank, "group": _default_group[0], "buffers": list(ipc_tensors)}
    return h


def dispose(fa: int) -> None:
    _handles.pop(fa, None)


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
    if out.data_ptr() != inp.data_ptr():
        out.copy_(inp)
    if dist.is_initialized() and dist.get_world_size(_handles[fa]["group"]) > 1:
        dist.all_reduce(out, group=_handles[fa]["group"])
