"""Host-side mirror of `flashinfer.moe_fused_gate` as the reference's router calls it
(python/sglang/srt/layers/moe/topk.py:709-733): DeepSeek-V3 biased grouped top-k in one HIP kernel
(csrc/moe_gate.hip).  No fallback: without the HIP library the import of fluent_mi355._lib fails."""
from __future__ import annotations

import ctypes

import torch

from ._lib import check, lib, stream_ptr

lib.fl_moe_fused_gate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p]
lib.fl_moe_fused_gate.restype = ctypes.c_int


def moe_fused_gate(input_tensor, bias, num_expert_group, topk_group, topk, num_fused_shared_experts=0,
                   routed_scaling_factor=1.0, apply_routed_scaling_factor_on_output=False, num_token_non_padded=None):
    """-> (topk_weights f32 [T, topk], topk_ids int32 [T, topk]).  `input_tensor` f32 [T, E] router logits, `bias` the
    e_score_correction_bias [E].  `num_token_non_padded` (optional int32 device scalar) folds the reference's
    _mask_topk_ids_padded_region post-process (topk.py:673-680, "TODO merge into kernel" at :725) into the kernel."""
    if num_fused_shared_experts:
        raise NotImplementedError("moe_fused_gate: num_fused_shared_experts > 0 is not supported "
                                  "(the reference routes that case to biased_grouped_topk_impl, topk.py:709)")
    if not input_tensor.is_cuda:
        raise RuntimeError("moe_fused_gate: input must be a CUDA/HIP tensor")
    x = input_tensor.to(torch.float32).contiguous()
    b = bias.to(device=x.device, dtype=torch.float32).contiguous()
    T, E = x.shape
    if b.numel() != E:
        raise RuntimeError(f"moe_fused_gate: bias has {b.numel()} entries for {E} experts")
    w = torch.empty(T, topk, dtype=torch.float32, device=x.device)
    ids = torch.empty(T, topk, dtype=torch.int32, device=x.device)
    if T == 0:
        return w, ids   # (empty tensors have no storage to point at)
    npad_t = None
    if num_token_non_padded is not None:
        npad_t = num_token_non_padded.to(device=x.device, dtype=torch.int32).contiguous()
    npad = npad_t.data_ptr() if npad_t is not None else 0
    check(lib.fl_moe_fused_gate(x.data_ptr(), b.data_ptr(), T, E, int(num_expert_group), int(topk_group), int(topk),
                                float(routed_scaling_factor), int(bool(apply_routed_scaling_factor_on_output)), npad,
                                w.data_ptr(), ids.data_ptr(), stream_ptr(x.device)), "fl_moe_fused_gate")
    return w, ids
