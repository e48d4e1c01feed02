"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

from conftest import PKG, ROOT


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        names |= set(re.findall(r"^\s*(?:const\s+char\s*\*|int)\s+(fl_[a-z0-9_]+)\s*\(", src, flags=re.M))
    return names


def test_library_exports_all_declared_symbols():
    so = os.path.join(PKG, "fluent_mi355", "libfluent_mi355.so")
    if not os.path.exists(so):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(so)
    decl = _declared_symbols()
    assert len(decl) >= 8
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
    assert not missing, missing
    lib.fl_version.restype = ctypes.c_int
    m = re.search(r"#define\s+FL_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "fluent_mi355.h")).read())
    assert lib.fl_version() == int(m.group(1)) == 102          # header, library and the ctypes binding agree
    from fluent_mi355 import _lib
    assert _lib.ABI_VERSION == 102
    # an argument struct of another layout is refused, not read past its end
    a = _lib.FlMlaDecodeArgs()
    a.struct_bytes = ctypes.sizeof(_lib.FlMlaDecodeArgs) - 8
    lib.fl_mla_decode.argtypes = [ctypes.POINTER(_lib.FlMlaDecodeArgs), ctypes.c_void_p]
    lib.fl_last_error.restype = ctypes.c_char_p
    assert lib.fl_mla_decode(ctypes.byref(a), None) == 1 and b"struct_bytes" in lib.fl_last_error()
    lib.fl_mla_num_parts.restype = ctypes.c_int
    # 64-row workgroups for every shape (mla_decode_fp8_y.hip) -> CUs / ceil(rows / 64) parts
    assert lib.fl_mla_num_parts(256, 128) == 128
    assert lib.fl_mla_num_parts(256, 256) == 64
    assert lib.fl_mla_num_parts(256, 64) == 256 and lib.fl_mla_num_parts(256, 16) == 256


def test_shim_modules_import_and_match_reference_names():
    import flash_mla_fp8
    import flash_mla_swap

    for n in ("get_mla_metadata", "flash_mla_with_kvcache", "flash_mla_ckv_fp8_per_token",
              "quantize_ckv_per_token_head", "quantize_and_cache_k", "dequantize_ckv_fused_indexed"):
        assert callable(getattr(flash_mla_fp8, n)), n
    for n in ("get_mla_metadata", "flash_mla_with_kvcache"):
        assert callable(getattr(flash_mla_swap, n)), n


def test_product_path_never_imports_oracle():
    for path in glob.glob(os.path.join(PKG, "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), path
