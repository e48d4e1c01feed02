"""CPU: the boundary as the reference calls it.  tests/golden/flashmla_backend_call_trace.json holds what
FlashMLABackend.forward_decode / the verify branch of forward_extend (flashmla_backend.py:88-256, exec'd from the reference
source by oracle/gen_golden.py:gen_call_trace) pass to `flash_mla_fp8` / `flash_mla_swap`: every recorded call must bind to
the signature of our drop-in function of the same module and name (keyword names verbatim)."""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "sglang-fluentllm_amd"))


def load_trace():
    with open(os.path.join(HERE, "golden", "flashmla_backend_call_trace.json")) as f:
        return json.load(f)["trace"]


def test_recorded_calls_bind_to_the_drop_in_signatures():
    import flash_mla_fp8
    import flash_mla_swap

    mods = {"flash_mla_fp8": flash_mla_fp8, "flash_mla_swap": flash_mla_swap}
    n = 0
    for case in load_trace():
        for call in case["calls"]:
            if call["module"] not in mods:
                continue   # (token_to_kv_pool.set_kv_buffer stays in the reference: it calls quantize_and_cache_k)
            fn = getattr(mods[call["module"]], call["fn"])
            inspect.signature(fn).bind(*[None] * len(call["args"]), **{k: None for k in call["kwargs"]})
            n += 1
    assert n >= 10


def test_module_dispatch_rule_of_the_trace():
    """get_flash_mla_module (flashmla_backend.py:18-22): M = s_q * H <= 56 and not per-token quantisation -> flash_mla_swap"""
    for case in load_trace():
        attend = [c for c in case["calls"] if c["fn"].startswith("flash_mla_")][0]
        M = case["s_q"] * case["H"]
        expect = "flash_mla_swap" if (M <= 56 and case["quant_method"] != "per_token_head") else "flash_mla_fp8"
        assert attend["module"] == expect, case["case"]


def test_construction_time_flashinfer_dependencies_exist_and_are_inert():
    """FlashInferMLAAttnBackend.__init__ (flashinfer_mla_backend.py:124-142) constructs these; decode / verify never run them."""
    import pytest
    import torch

    import flashinfer
    from flashinfer.comm import vllm_ar

    ws = torch.empty(16, dtype=torch.uint8)
    ragged = flashinfer.BatchPrefillWithRaggedKVCacheWrapper(ws, "NHD")
    paged = flashinfer.BatchMLAPagedAttentionWrapper(ws, backend="auto")
    graph = flashinfer.BatchMLAPagedAttentionWrapper(ws, use_cuda_graph=True, qo_indptr=torch.zeros(3), kv_indptr=torch.zeros(3),
                                                     kv_indices=torch.zeros(8), kv_len_arr=torch.zeros(2), backend="auto")
    assert graph._use_cuda_graph and graph._kv_indices_buf is not None and not paged._use_cuda_graph
    paged.plan(torch.zeros(3), torch.zeros(3), torch.zeros(8), torch.zeros(2), 16, 512, 64, 64, False, 0.1, torch.bfloat16, torch.bfloat16)
    ragged.begin_forward(torch.zeros(3), torch.zeros(3), num_qo_heads=16, num_kv_heads=1, head_dim_qk=192)
    assert paged.last_plan is not None
    with pytest.raises(RuntimeError):
        paged.run(None, None, None, None)
    with pytest.raises(RuntimeError):
        ragged.forward(None, None, None)
    # C4 entry points (sglang/srt/_custom_ops.py:21-60) — world size 1: out = inp
    h = vllm_ar.init_custom_ar([0], torch.zeros(1), 0, True)
    x, y = torch.arange(6.0), torch.zeros(6)
    vllm_ar.all_reduce(h, x, y, 0, 0)
    assert torch.equal(x, y) and vllm_ar.meta_size() == 0
    vllm_ar.dispose(h)
