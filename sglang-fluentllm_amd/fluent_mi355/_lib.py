import ctypes
import os

import torch

LIB_PATH = os.environ.get("FLUENT_MI355_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                            "libfluent_mi355.so")

ABI_VERSION = 102   # include/fluent_mi355.h: FL_ABI_VERSION
_c_void_p = ctypes.c_void_p
_i32p = ctypes.c_void_p
_f32p = ctypes.c_void_p


class FlMlaDecodeArgs(ctypes.Structure):
    """Mirror of `struct FlMlaDecodeArgs` in include/fluent_mi355.h (field order is ABI)."""

    _fields_ = [
        ("struct_bytes", ctypes.c_int32), ("kv_format", ctypes.c_int32), ("bs", ctypes.c_int32), ("s_q", ctypes.c_int32), ("h_q", ctypes.c_int32),
        ("d_nope", ctypes.c_int32), ("d_rope", ctypes.c_int32), ("causal", ctypes.c_int32),
        ("num_parts", ctypes.c_int32),
        ("softmax_scale", ctypes.c_float), ("descale_q", _f32p), ("descale_k", _f32p),
        ("q_nope", _c_void_p), ("q_rope", _c_void_p), ("q_scale", _f32p),
        ("k_nope", _c_void_p), ("k_rope", _c_void_p), ("k_scale", _f32p),
        ("num_pages", ctypes.c_int64),
        ("block_table", _i32p), ("block_table_stride", ctypes.c_int64),
        ("cache_seqlens", _i32p), ("tile_scheduler_metadata", _i32p), ("num_splits", _i32p),
        ("out", _c_void_p), ("lse", _f32p), ("o_accum", _f32p), ("lse_accum", _f32p),
        ("q_bf16", _c_void_p), ("block_table_cols", ctypes.c_int64),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"fluent_mi355: {LIB_PATH} is missing — build it with `make -C sglang-fluentllm_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU/PyTorch fallback for this path.")
    lib_ = ctypes.CDLL(LIB_PATH)
    lib_.fl_last_error.restype = ctypes.c_char_p
    lib_.fl_version.restype = ctypes.c_int
    if lib_.fl_version() != ABI_VERSION:
        raise RuntimeError(f"fluent_mi355: {LIB_PATH} reports ABI {lib_.fl_version()}, this binding was written for {ABI_VERSION} "
                           "(include/fluent_mi355.h: FL_ABI_VERSION) — rebuild the library")
    lib_.fl_device_cu_count.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib_.fl_mla_num_parts.argtypes = [ctypes.c_int, ctypes.c_int]
    lib_.fl_mla_get_metadata.argtypes = [_i32p, ctypes.c_int, ctypes.c_int, _i32p, _i32p, _c_void_p]
    lib_.fl_mla_quant_q.argtypes = [_c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_void_p, _f32p,
                                    _c_void_p, _c_void_p]
    lib_.fl_mla_quant_store_k.argtypes = [_c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _i32p, _c_void_p,
                                          _f32p, _c_void_p, ctypes.c_int64, _c_void_p]
    lib_.fl_mla_quant_q_store_k.argtypes = [_c_void_p, ctypes.c_int64, _i32p, _c_void_p, _f32p, _c_void_p, ctypes.c_int64,
                                            _c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_void_p, _f32p, _c_void_p,
                                            _c_void_p]
    lib_.fl_mla_dequant_gather.argtypes = [_c_void_p, _c_void_p, _f32p, _i32p, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int64, _c_void_p, _c_void_p, _c_void_p]
    lib_.fl_mla_decode.argtypes = [ctypes.POINTER(FlMlaDecodeArgs), _c_void_p]
    for name in ("fl_device_cu_count", "fl_mla_num_parts", "fl_mla_get_metadata", "fl_mla_quant_q",
                 "fl_mla_quant_store_k", "fl_mla_quant_q_store_k", "fl_mla_dequant_gather", "fl_mla_decode"):
        getattr(lib_, name).restype = ctypes.c_int
    return lib_


lib = _load()


def check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"{what} failed (status {status}): {lib.fl_last_error().decode()}")


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's CURRENT stream (what the reference's ops launch on; graph-capturable)."""
    return torch.cuda.current_stream(device).cuda_stream


_cu_cache = {}


def cu_count(device) -> int:
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _cu_cache:
        v = ctypes.c_int(0)
        check(lib.fl_device_cu_count(idx, ctypes.byref(v)), "fl_device_cu_count")
        _cu_cache[idx] = v.value
    return _cu_cache[idx]
