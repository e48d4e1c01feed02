"""Host-side mirror of the `deep_gemm` / `flashinfer.quantization` / `flashinfer.activation` / `eps.executor`
operator API on the MoE + quant-linear path.  Same names and argument conventions as the reference's call sites
(python/sglang/srt/layers/moe/gemms/fp8/fire.py:18; moe/executors/fp8_eps_executor.py:52-78;
moe/executors/deep_ep_executor.py:583-586,607-613,655-688; dense/gemms/fp8/deep_geem.py:55;
dense/gemms/fp8/fp8_kernel.py:430-462).  Outputs are caller-allocated where the reference allocates them
(fp8_eps_executor.py:46-51,68-73).  Marshals data_ptr()+stream into the C-ABI; nothing here computes."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from ._lib import check, lib, stream_ptr

_ONE_BYTE = (torch.uint8, torch.int8, torch.float8_e4m3fn)
MODE_OFFSET, MODE_CONTIGUOUS, MODE_MASKED, MODE_DENSE = 0, 1, 2, 3


class FlGemmArgs(ctypes.Structure):
    """Mirror of `struct FlGemmArgs` in include/fluent_mi355.h."""

    _fields_ = [
        ("mode", ctypes.c_int32), ("num_groups", ctypes.c_int32), ("M", ctypes.c_int64),
        ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("rows_per_group", ctypes.c_int64), ("expected_m", ctypes.c_int64),
        ("A", ctypes.c_void_p), ("As", ctypes.c_void_p),
        ("as_stride_m", ctypes.c_int64), ("as_stride_k", ctypes.c_int64), ("as_stride_g", ctypes.c_int64),
        ("W", ctypes.c_void_p), ("Ws", ctypes.c_void_p), ("out", ctypes.c_void_p), ("group_meta", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
    ]


lib.fl_grouped_gemm_fp8.argtypes = [ctypes.POINTER(FlGemmArgs), ctypes.c_void_p]
lib.fl_grouped_gemm_fp8.restype = ctypes.c_int
lib.fl_gemm_set_num_cus.argtypes = [ctypes.c_int]
lib.fl_gemm_get_num_cus.restype = ctypes.c_int
lib.fl_quant_1x128.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
lib.fl_quant_1x128.restype = ctypes.c_int
lib.fl_silu_and_mul.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
lib.fl_silu_and_mul.restype = ctypes.c_int


def _req(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def ceil_div(a: int, b: int) -> int:
    return (a + b - 1) // b


def get_num_sms() -> int:
    """deep_gemm.get_num_sms: the cap in force, or the device's CU count when none was set (tbo_executor.py:129-134 saves this
    value and restores it)."""
    return lib.fl_gemm_get_num_cus()


def set_num_sms(n: int) -> None:
    """deep_gemm.set_num_sms (tbo/tbo_executor.py:129-134): later grouped / dense GEMM launches take at most `n` workgroup
    slots (a persistent walk over the tile list), leaving the rest of the chip to the kernels of the other stream.  A
    value >= the device's CU count (what get_num_sms reports by default) removes the cap."""
    n = int(n)
    check(lib.fl_gemm_set_num_cus(0 if n >= _device_cus() else n), "fl_gemm_set_num_cus")


def _device_cus() -> int:
    from ._lib import cu_count
    return cu_count(torch.device("cuda", torch.cuda.current_device())) if torch.cuda.is_available() else 1 << 30


def get_col_major_tma_aligned_tensor(x: torch.Tensor) -> torch.Tensor:
    """Scales layout helper of the reference (column-major, rows aligned to 4): our kernels take arbitrary strides,
    so this only has to return a tensor with the same values."""
    return x


_SPLITK_MAX_ROWS = 1024
_SPLITK_BYTES = 64 << 20
_splitk_ws = {}


def _splitk_workspace(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_ptr(device))
    ws = _splitk_ws.get(key)
    if ws is None:
        ws = _splitk_ws[key] = torch.empty(_SPLITK_BYTES, dtype=torch.uint8, device=device)
    return ws


def _gemm(mode, lhs, rhs, out, meta, rows_per_group=0, expected_m=0):
    A, As = lhs
    W, Ws = rhs
    for t, n in ((A, "lhs"), (As, "lhs scales"), (W, "rhs"), (Ws, "rhs scales"), (out, "out")):
        _req(t.is_cuda, f"{n} must be a device tensor (no CPU fallback on this path)")
    _req(A.dtype in _ONE_BYTE and W.dtype in _ONE_BYTE, "lhs / rhs must be fp8 (e4m3)")
    _req(As.dtype == torch.float32 and Ws.dtype == torch.float32, "scales must be float32")
    _req(out.dtype == torch.bfloat16 and out.is_contiguous(), "out must be contiguous bfloat16")
    _req(A.is_contiguous() and W.is_contiguous() and Ws.is_contiguous(), "lhs, rhs and rhs scales must be contiguous")
    a = FlGemmArgs()
    a.mode = mode
    K = A.shape[-1]
    if W.dim() == 2:
        E, (N, Kw) = 1, W.shape
    else:
        E, N, Kw = W.shape
    _req(K == Kw, f"K mismatch: lhs {K} vs rhs {Kw}")
    _req(out.shape[-1] == N, "out / rhs N mismatch")
    _req(tuple(Ws.shape[-2:]) == (ceil_div(N, 128), K // 128), f"rhs scales must be [.., {ceil_div(N,128)}, {K//128}]")
    a.num_groups, a.N, a.K = E, N, K
    a.M = A.numel() // K
    _req(out.numel() == a.M * N, "out / lhs row mismatch")
    _req(As.shape[-1] == K // 128 and As.numel() // (K // 128) >= 0, "lhs scales must be [.., K/128]")
    a.rows_per_group = rows_per_group
    a.expected_m = int(expected_m)
    a.A, a.As, a.W, a.Ws, a.out = A.data_ptr(), As.data_ptr(), W.data_ptr(), Ws.data_ptr(), out.data_ptr()
    if mode == MODE_MASKED:
        _req(As.dim() == 3, "masked: lhs scales must be [G, M, K/128]")
        a.as_stride_g, a.as_stride_m, a.as_stride_k = As.stride(0), As.stride(1), As.stride(2)
    else:
        _req(As.dim() == 2 and As.shape[0] >= a.M, "lhs scales must be [M, K/128]")
        a.as_stride_g, a.as_stride_m, a.as_stride_k = 0, As.stride(0), As.stride(1)
    if meta is not None:
        _req(meta.is_cuda and meta.dtype == torch.int32 and meta.is_contiguous(), "group metadata must be int32 on device")
        a.group_meta = meta.data_ptr()
    if mode == MODE_DENSE and a.M <= _SPLITK_MAX_ROWS:
        # few output tiles (decode projections): scratch for the library's split-K, one buffer per (device, stream)
        ws = _splitk_workspace(A.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    check(lib.fl_grouped_gemm_fp8(ctypes.byref(a), stream_ptr(A.device)), "fl_grouped_gemm_fp8")


def m_grouped_gemm_fp8_fp8_bf16_nt_offset(lhs, rhs, out, exclusive_sum, use_pdl: bool = False) -> None:
    """out[ex[e]:ex[e+1]] = lhs[ex[e]:ex[e+1]] @ rhs[e]^T with 1x128 / 128x128 block scales (fire.py:18)."""
    _req(exclusive_sum.numel() == rhs[0].shape[0] + 1, "exclusive_sum must have num_groups + 1 entries")
    _gemm(MODE_OFFSET, lhs, rhs, out, exclusive_sum)


def m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(lhs, rhs, out, m_indices, pdl: bool = False) -> None:
    """Row i uses group m_indices[i]; groups are 128-row aligned (deep_ep_executor.py:583-586,607-613)."""
    _req(m_indices.numel() == lhs[0].shape[0], "m_indices must have one entry per row")
    _gemm(MODE_CONTIGUOUS, lhs, rhs, out, m_indices)


def m_grouped_gemm_fp8_fp8_bf16_nt_masked(lhs, rhs, out, masked_m, expected_m, pdl: bool = False) -> None:
    """lhs [G, M, K]: only the first masked_m[g] rows of group g are computed (deep_ep_executor.py:655-662,681-688)."""
    _req(lhs[0].dim() == 3 and masked_m.numel() == lhs[0].shape[0], "masked: lhs must be [G, M, K], masked_m [G]")
    _gemm(MODE_MASKED, lhs, rhs, out, masked_m, rows_per_group=lhs[0].shape[1], expected_m=expected_m)


def gemm_fp8_fp8_bf16_nt(lhs, rhs, out, pdl: bool = False) -> None:
    """Dense block-scaled GEMM out = lhs @ rhs^T (deep_geem.py:55)."""
    _gemm(MODE_DENSE, lhs, rhs, out, None)


# ---------------------------------------------------------------------------------------------- quantisation
def _quant(x, x_q, x_s, eps):
    _req(x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(), "x must be contiguous bfloat16 on device")
    _req(x_q.is_contiguous() and x_q.dtype in _ONE_BYTE and x_s.dtype == torch.float32, "bad output dtypes")
    K = x.shape[-1]
    M = x.numel() // K
    _req(x_s.dim() == 2 and x_s.shape[1] == K // 128 and x_s.shape[0] >= M, "scales must be [M, K/128] (any strides)")
    check(lib.fl_quant_1x128(x.data_ptr(), M, K, eps, x_q.data_ptr(), x_s.data_ptr(), x_s.stride(0), x_s.stride(1),
                             stream_ptr(x.device)), "fl_quant_1x128")


def quant_1x128(x, out_fp8, out_scale, exclusive_sum, num_groups, max_shape_m, max_shape_m_padded, K) -> None:
    """flashinfer.quantization.quant_1x128 (fp8_eps_executor.py:53-55,75-77): per-token 1x128 quantisation of the
    routed rows; scale (m, kb) is written at out_scale[m, kb] (the strided [M, K/128] view the caller built)."""
    _quant(x, out_fp8, out_scale, 1e-10)


def sgl_per_token_group_quant_fp8(x, x_q, x_s, group_size, eps, fp8_min, fp8_max, scale_ue8m0=False) -> None:
    """flashinfer.sgl_per_token_group_quant_fp8 (fp8_kernel.py:457-460)."""
    _req(group_size == 128 and not scale_ue8m0, "only group_size=128 float scales")
    s2 = x_s if x_s.dim() == 2 else x_s.reshape(-1, x_s.shape[-1])
    _quant(x, x_q, s2, float(eps))


def per_token_group_quant_fp8(x, group_size=128, eps=1e-10, column_major_scales=False):
    """Convenience: allocate + quantise -> (x_q fp8 [M,K], x_s f32 [M,K/128]) (fp8_kernel.py:430-462 semantics)."""
    K = x.shape[-1]
    M = x.numel() // K
    x_q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    if column_major_scales:
        x_s = torch.empty((K // 128, (M + 3) // 4 * 4), dtype=torch.float32, device=x.device).permute(-1, -2)[:M, :]
    else:
        x_s = torch.empty((M, K // 128), dtype=torch.float32, device=x.device)
    _quant(x.reshape(M, K), x_q, x_s, eps)
    return x_q, x_s


# ---------------------------------------------------------------------------------------------- activation
def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """flashinfer.silu_and_mul(x, out): silu(x[..., :I]) * x[..., I:] (activation.py:58-60)."""
    _req(x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(), "x must be contiguous bfloat16 on device")
    I = x.shape[-1] // 2
    M = x.numel() // (2 * I)
    if out is None:
        out = torch.empty(x.shape[:-1] + (I,), dtype=torch.bfloat16, device=x.device)
    check(lib.fl_silu_and_mul(x.data_ptr(), M, I, out.data_ptr(), None, None, 0, 0, stream_ptr(x.device)),
          "fl_silu_and_mul")
    return out


def silu(gate_up: torch.Tensor, exclusive_sum: torch.Tensor, num_tokens_hint: int) -> torch.Tensor:
    """eps.executor.silu (fp8_eps_executor.py:62): rows past exclusive_sum[-1] are don't-care; all rows are computed."""
    return silu_and_mul(gate_up)


def silu_and_mul_fuse_block_quant(x, scale, out, enable_pdl=True, masked_m=None, expected_m=None, num_experts=None):
    """flashinfer.activation.silu_and_mul_fuse_block_quant (activation.py:73, deep_ep_executor.py:676):
    out fp8 [.., I] = quant_1x128(silu(x[..,:I]) * x[..,I:]), scale f32 [.., I/128] (any strides) -> (out, scale)."""
    _req(x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(), "x must be contiguous bfloat16 on device")
    I = x.shape[-1] // 2
    if x.dim() == 3 and masked_m is not None:
        silu_and_mul_masked_post_quant_fwd(x, out, scale, 128, masked_m)
    elif x.dim() == 3:
        G, M, _ = x.shape
        for g in range(G):   # per-group scale strides may differ from a flat view; G is small (local experts)
            check(lib.fl_silu_and_mul(x[g].data_ptr(), M, I, None, out[g].data_ptr(), scale[g].data_ptr(),
                                      scale[g].stride(0), scale[g].stride(1), stream_ptr(x.device)), "fl_silu_and_mul")
    else:
        M = x.numel() // (2 * I)
        check(lib.fl_silu_and_mul(x.data_ptr(), M, I, None, out.data_ptr(), scale.data_ptr(), scale.stride(0),
                                  scale.stride(1), stream_ptr(x.device)), "fl_silu_and_mul")
    return out, scale


def silu_and_mul_masked_post_quant_fwd(input, output, output_scale, quant_group_size, masked_m) -> None:
    """Same signature and effect as the reference's Triton launcher (deep_ep_executor.py:106-170): input bf16
    [G, M, 2I] contiguous, output fp8 [G, M, I] contiguous, output_scale f32 [G, M, I/128] (any strides), masked_m int32 [G];
    rows >= masked_m[g] are neither read nor written.  One launch."""
    _req(input.is_cuda and input.dtype == torch.bfloat16 and input.is_contiguous() and input.dim() == 3,
         "input must be contiguous bfloat16 [G, M, 2I] on device")
    _req(output.dtype == torch.float8_e4m3fn and output.is_contiguous(), "output must be contiguous float8_e4m3fn")
    _req(int(quant_group_size) == 128, "quant_group_size must be 128")
    _req(output_scale.dtype == torch.float32 and output_scale.dim() == 3, "output_scale must be float32 [G, M, I/128]")
    G, M, two_i = input.shape
    _req(masked_m.numel() == G, "masked_m must have one entry per group")
    mm = masked_m if masked_m.dtype == torch.int32 else masked_m.to(torch.int32)
    lib.fl_silu_and_mul_masked.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_void_p]
    lib.fl_silu_and_mul_masked.restype = ctypes.c_int
    check(lib.fl_silu_and_mul_masked(input.data_ptr(), G, M, two_i // 2, mm.contiguous().data_ptr(), output.data_ptr(),
                                     output_scale.data_ptr(), output_scale.stride(0), output_scale.stride(1),
                                     output_scale.stride(2), stream_ptr(input.device)), "fl_silu_and_mul_masked")
