"""B1: the small dense bf16 GEMMs between the big kernels of a DeepSeek decode layer, hand-written on MFMA
(csrc/bmm_bf16.hip) — the weight-absorption `torch.bmm`s of DeepseekV2AttentionMLA.forward_absorb
(srt/models/deepseek_v2.py:840, :886; weights stored k-contiguous, :1632-1633) and the router GEMM
`flashinfer.dsv3_router_gemm` (:177-179).  `bmm(a, b, out=...)` has torch.bmm's meaning for the operand layouts those call
sites use (a[..., k] and b[:, k, :] contiguous in k); anything else raises — there is no library fallback."""
from __future__ import annotations

import ctypes

import torch

from ._lib import check, lib, stream_ptr

_vp, _i64, _i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.fl_bmm_bf16_nt.argtypes = [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]
lib.fl_bmm_bf16_nt.restype = _i32


def bmm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None, out_dtype: torch.dtype | None = None) -> torch.Tensor:
    """out[i] = a[i] @ b[i]:  a [B, M, K] bf16 with stride(2) == 1, b [B, K, N] bf16 with stride(1) == 1 (k-contiguous, e.g.
    `w.transpose(1, 2)` of a contiguous [B, N, K]), out [B, M, N] bf16 or f32 with stride(2) == 1 (any outer strides: a view
    into a larger tensor is fine, as in `torch.bmm(..., out=Q[..., :512].transpose(0, 1))`).  fp32 accumulation."""
    if a.dim() != 3 or b.dim() != 3 or a.shape[0] != b.shape[0] or a.shape[2] != b.shape[1]:
        raise RuntimeError(f"bmm: shapes {tuple(a.shape)} x {tuple(b.shape)}")
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or not a.is_cuda or b.device != a.device:
        raise RuntimeError("bmm: bf16 CUDA/HIP operands on one device")
    B, M, K = a.shape
    N = b.shape[2]
    if M and K and (a.stride(2) != 1 or b.stride(1) != 1):
        raise RuntimeError("bmm: both operands must be contiguous along k (a.stride(2) == 1, b.stride(1) == 1)")
    if out is None:
        out = torch.empty(B, M, N, dtype=out_dtype or torch.bfloat16, device=a.device)
    if out.shape != (B, M, N) or out.dtype not in (torch.bfloat16, torch.float32) or (M and N and out.stride(2) != 1):
        raise RuntimeError("bmm: out must be [B, M, N] bf16 / f32 with a contiguous last dimension")
    if B == 0 or M == 0:
        return out
    check(lib.fl_bmm_bf16_nt(a.data_ptr(), b.data_ptr(), out.data_ptr(), B, M, N, K, a.stride(0), a.stride(1), b.stride(0),
                             b.stride(2), out.stride(0), out.stride(1), int(out.dtype == torch.float32), stream_ptr(a.device)),
          "fl_bmm_bf16_nt")
    return out


lib.fl_gemm_bf16_nt_splitk_workspace_bytes.argtypes = [_i64, _i32, _i32]
lib.fl_gemm_bf16_nt_splitk_workspace_bytes.restype = _i64
lib.fl_gemm_bf16_nt_splitk.argtypes = [_vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i64, _i32, _vp, _i64, _vp]
lib.fl_gemm_bf16_nt_splitk.restype = _i32
_SPLITK_WS_MAX = 64 << 20
_splitk_ws = {}


def _router_workspace(device, nbytes):
    """f32 partials of the split-K router GEMM.  Eager calls share one buffer per (device, stream), grown on demand.  A call made while
    the stream is CAPTURING gets a fresh buffer every time: it comes out of the graph's private pool and the graph owns it — nothing
    allocated under one capture is handed to another capture (another pool; the first graph may be gone) or replaced while earlier
    captured kernels still point at it."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_ptr(device))
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _splitk_ws[key] = torch.empty(max(nbytes, 8 << 20), dtype=torch.uint8, device=device)
    return ws


def dsv3_router_gemm(hidden_states: torch.Tensor, weight: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """router logits hidden [T, K] x weight [E, K]^T (models/deepseek_v2.py:177-179) -> [T, E] in `out_dtype`"""
    if hidden_states.dim() != 2 or weight.dim() != 2 or hidden_states.shape[1] != weight.shape[1]:
        raise RuntimeError("dsv3_router_gemm: hidden [T, K], weight [E, K]")
    T, K = hidden_states.shape
    E = weight.shape[0]
    if (hidden_states.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and hidden_states.is_cuda and T > 0 and E % 64 == 0
            and K % 64 == 0 and hidden_states.stride(1) == 1 and weight.stride(1) == 1 and out_dtype in (torch.float32, torch.bfloat16)):
        need = int(lib.fl_gemm_bf16_nt_splitk_workspace_bytes(T, E, K))
        if 0 < need <= _SPLITK_WS_MAX and hidden_states.stride(0) % 8 == 0 and weight.stride(0) % 8 == 0:
            # few output tiles, long K: k split across workgroups + a deterministic reduce (csrc/bmm_bf16.hip, B3)
            ws = _router_workspace(hidden_states.device, need)
            out = torch.empty(T, E, dtype=out_dtype, device=hidden_states.device)
            check(lib.fl_gemm_bf16_nt_splitk(hidden_states.data_ptr(), weight.data_ptr(), out.data_ptr(), T, E, K, hidden_states.stride(0),
                                             weight.stride(0), out.stride(0), int(out_dtype == torch.float32), ws.data_ptr(), ws.numel(),
                                             stream_ptr(hidden_states.device)), "fl_gemm_bf16_nt_splitk")
            return out
    o = bmm(hidden_states.unsqueeze(0), weight.unsqueeze(0).transpose(1, 2),
            out_dtype=torch.float32 if out_dtype == torch.float32 else torch.bfloat16)[0]
    return o if o.dtype == out_dtype else o.to(out_dtype)
