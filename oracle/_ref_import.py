"""TEST INFRASTRUCTURE ONLY (container-side): import harness for the real reference.

Makes `/root/reference/python/sglang` importable on CPU in THIS container so that
`oracle/gen_golden.py` can run the reference's own torch-native code and emit golden
vectors.  Nothing here travels to the GPU box in a usable form: /root/reference does
not exist there, and no test, smoke() or bench reads it at run time.

Recipe follows SURVEY.md §8(c): stub absent third-party packages with MagicMock,
force the non-HIP branches (`torch.version.hip = None`), never write bytecode into
the read-only reference tree.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = "/root/reference/python"

_STUB_TOPLEVEL = {
    "pybase64", "zmq", "eps", "flashinfer", "flash_mla_fp8", "flash_mla_swap", "flash_mla",
    "deep_gemm", "deep_gemm_oss", "compressed_tensors", "openai", "vllm", "orjson", "uvloop",
    "sgl_kernel", "outlines", "xgrammar", "llguidance", "interegular", "partial_json_parser",
    "setproctitle", "torchao", "decord", "cuda", "pynvml", "modelscope", "torch_memory_saver",
    "mooncake", "nixl", "deep_ep", "fast_hadamard_transform", "tilelang", "blobfile", "tiktoken",
    "IPython", "gguf", "einx", "msgspec", "soundfile", "scipy_stub", "triton_kernels",
}


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_TOPLEVEL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


def install():
    """Idempotently prepare sys.path / stubs. Returns True when the reference is present."""
    if not os.path.isdir(REF_ROOT):
        return False
    sys.dont_write_bytecode = True
    os.environ.setdefault("PYTHONPYCACHEPREFIX", "/tmp/pyc")
    sys.pycache_prefix = "/tmp/pyc"
    import torch

    torch.version.hip = None
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return True


def load_functions_from_source(path, names, namespace):
    """exec() selected top-level FunctionDefs of a reference file that does not import as a
    module (stale imports) into `namespace`. Used only to RUN them here."""
    import ast

    with open(path) as f:
        tree = ast.parse(f.read())
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=picked, type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return namespace


def load_method_from_source(path, cls_name, method_name, namespace):
    """exec() one method of one class of a reference file (whose module does not import under
    stubs) as a free function in `namespace`; returns it. Used only to RUN it here."""
    import ast

    with open(path) as f:
        tree = ast.parse(f.read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls_name:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == method_name:
                    exec(compile(ast.Module(body=[item], type_ignores=[]), path, "exec"), namespace)
                    return namespace[method_name]
    raise KeyError(f"{cls_name}.{method_name} not found in {path}")


def load_function_from_source(path, func_name, namespace):
    """exec() one module-level function of a reference file (whose module does not import under stubs) in `namespace`,
    WITHOUT its decorators (e.g. @torch.compile: the eager function is the statement we want); returns it."""
    import ast

    with open(path) as f:
        tree = ast.parse(f.read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == func_name:
            node.decorator_list = []
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
            return namespace[func_name]
    raise KeyError(f"{func_name} not found in {path}")


def load_triton_functions(path, names):
    """exec() module-level functions of a reference file WITH their decorators (@triton.jit kernels and their Python
    launchers) and run them on the CPU through Triton's interpreter (TRITON_INTERPRET=1 must be set before triton is
    imported).  The stock interpreter's range()/tl.range() yield Python ints, on which the reference's kernels call
    `.to(tl.int64)`: the namespace carries a range() and a `tl` proxy whose loop variables are interpreter tensors.
    Programs of a grid run one after the other, so atomics hand out positions in program order."""
    import ast
    import builtins
    import os

    assert os.environ.get("TRITON_INTERPRET") == "1", "set TRITON_INTERPRET=1 before importing triton"
    import torch
    import triton
    import triton.language as tl

    def as_int(v):
        return int(v.handle.data.item()) if hasattr(v, "handle") else int(v)

    def py_range(*a):
        try:
            zero = tl.sum(tl.full([1], 0, tl.int32), 0)   # inside a kernel: an interpreter scalar
        except Exception:
            zero = None                                   # host code of a launcher
        for v in builtins.range(*[as_int(v) for v in a]):
            yield (zero + v) if zero is not None else v

    class TL:
        def __getattr__(self, k):
            return getattr(tl, k)

        @staticmethod
        def range(a, b=None, step=None, **kw):
            one = tl.full([1], 0, tl.int32)
            for v in builtins.range(*[as_int(v) for v in (a, b, step) if v is not None]):
                yield tl.sum(one, 0) + v

    ns = {"torch": torch, "triton": triton, "tl": TL(), "_tl_module": tl, "range": py_range}
    with open(path) as f:
        tree = ast.parse(f.read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    missing = [n for n in names if n not in ns]
    if missing:
        raise KeyError(f"{missing} not found in {path}")
    return ns
