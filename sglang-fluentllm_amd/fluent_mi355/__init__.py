"""fluent_mi355 — ctypes binding of libfluent_mi355.so (the C-ABI in include/fluent_mi355.h).

Host-side plumbing only: tensors are torch (device memory, streams); every compute call goes
through the C-ABI to hand-written gfx950 HIP kernels.  There is NO CPU or eager-PyTorch
fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from ._lib import lib, check, stream_ptr, cu_count, FlMlaDecodeArgs, LIB_PATH  # noqa: F401
