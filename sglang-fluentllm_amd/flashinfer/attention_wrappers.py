"""Construction-time dependencies of `FlashMLABackend(model_runner)`: its base class
`FlashInferMLAAttnBackend.__init__` (python/sglang/srt/layers/attention/flashinfer_mla_backend.py:124-142) builds
`flashinfer.BatchPrefillWithRaggedKVCacheWrapper(workspace, "NHD")` and three `flashinfer.BatchMLAPagedAttentionWrapper(
workspace, backend="auto")`, and its indices updaters call `.plan(...)` / `.begin_forward(...)` on them
(:344, :622-650).  On the FP8 MLA decode / verify path every attention call goes to `flash_mla_fp8` / `flash_mla_swap`
instead (flashmla_backend.py:105-256), so these wrappers are INERT here: they hold the workspace, record the last plan, and
refuse to run — a prefill through them is a different attention family (out of scope) and must fail loudly,
not silently fall back."""
from __future__ import annotations


class _InertWrapper:
    _what = "flashinfer attention wrapper"

    def __init__(self, float_workspace_buffer=None, *args, use_cuda_graph: bool = False, **kwargs):
        self._float_workspace_buffer = float_workspace_buffer
        self._use_cuda_graph = bool(use_cuda_graph)
        # attributes the reference's fast_mla_decode_plan / graph paths read or assign (flashinfer_mla_backend.py:760-820)
        self._kv_indices_buf = kwargs.get("kv_indices")
        self._kv_indptr_buf = kwargs.get("kv_indptr")
        self._qo_indptr_buf = kwargs.get("qo_indptr")
        self._kv_len_arr_buf = kwargs.get("kv_len_arr")
        self._init_args, self._init_kwargs = args, kwargs
        self.last_plan = None

    def plan(self, *args, **kwargs):
        self.last_plan = (args, kwargs)

    begin_forward = plan

    def end_forward(self):
        self.last_plan = None

    def run(self, *args, **kwargs):
        raise RuntimeError(f"{self._what}.run: not on the FP8 MLA decode / grouped-GEMM hot path this library replaces "
                           "(flashmla_backend.py routes decode and verify to flash_mla_fp8 / flash_mla_swap); prefill through "
                           "flashinfer has no MI355X implementation here")

    forward = run
    forward_return_lse = run
    run_return_lse = run


class BatchPrefillWithRaggedKVCacheWrapper(_InertWrapper):
    """flashinfer.BatchPrefillWithRaggedKVCacheWrapper(workspace, kv_layout) — flashinfer_mla_backend.py:124-126, used by
    forward_extend's ragged prefill (:449)."""
    _what = "BatchPrefillWithRaggedKVCacheWrapper"

    def __init__(self, float_workspace_buffer=None, kv_layout: str = "NHD", *args, **kwargs):
        super().__init__(float_workspace_buffer, *args, **kwargs)
        self._kv_layout = kv_layout


class BatchMLAPagedAttentionWrapper(_InertWrapper):
    """flashinfer.BatchMLAPagedAttentionWrapper(workspace, backend="auto", use_cuda_graph=..., qo_indptr=..., kv_indptr=...,
    kv_indices=..., kv_len_arr=...) — flashinfer_mla_backend.py:128-142 and the cuda-graph capture paths (:196-260)."""
    _what = "BatchMLAPagedAttentionWrapper"

    def __init__(self, float_workspace_buffer=None, use_cuda_graph: bool = False, qo_indptr=None, kv_indptr=None, kv_indices=None,
                 kv_len_arr=None, backend: str = "auto", **kwargs):
        super().__init__(float_workspace_buffer, use_cuda_graph=use_cuda_graph, qo_indptr=qo_indptr, kv_indptr=kv_indptr,
                         kv_indices=kv_indices, kv_len_arr=kv_len_arr, **kwargs)
        self._backend = backend
