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


# ---- R1b: the plain top-k routers of srt/layers/moe/topk.py (csrc/moe_gate.hip: topk_gate_kernel) ----
lib.fl_topk_gate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.fl_topk_gate.restype = ctypes.c_int


def _topk_gate(logits, bias, topk_weights, topk_ids, score_fn, renormalize, scale, what):
    """Runs fl_topk_gate INTO the caller's output tensors (the reference allocates them and passes them in)."""
    if not logits.is_cuda:
        raise RuntimeError(f"{what}: gating_output must be a CUDA/HIP tensor")
    if logits.dim() != 2 or topk_weights.dim() != 2 or topk_weights.shape != topk_ids.shape or topk_weights.shape[0] != logits.shape[0]:
        raise RuntimeError(f"{what}: expected gating_output [T, E] and outputs [T, topk] (got {tuple(logits.shape)}, "
                           f"{tuple(topk_weights.shape)}, {tuple(topk_ids.shape)})")
    if topk_weights.dtype != torch.float32 or topk_ids.dtype not in (torch.int32, torch.int64):
        raise RuntimeError(f"{what}: topk_weights must be float32 and topk_ids int32 / int64")
    T, E = logits.shape
    K = topk_weights.shape[1]
    if T == 0:
        return
    x = logits.to(torch.float32).contiguous()
    b = None if bias is None else bias.to(device=x.device, dtype=torch.float32).contiguous()
    if b is not None and b.numel() != E:
        raise RuntimeError(f"{what}: bias has {b.numel()} entries for {E} experts")
    w = topk_weights if topk_weights.is_contiguous() else torch.empty(T, K, dtype=torch.float32, device=x.device)
    ids = topk_ids if (topk_ids.dtype == torch.int32 and topk_ids.is_contiguous()) else torch.empty(T, K, dtype=torch.int32, device=x.device)
    check(lib.fl_topk_gate(x.data_ptr(), 0 if b is None else b.data_ptr(), T, E, K, int(score_fn), int(bool(renormalize)), float(scale),
                           w.data_ptr(), ids.data_ptr(), stream_ptr(x.device)), "fl_topk_gate")
    if w is not topk_weights:
        topk_weights.copy_(w)
    if ids is not topk_ids:
        topk_ids.copy_(ids)


def topk_softmax(topk_weights, topk_ids, gating_output, renormalize=False):
    """flashinfer.topk_softmax as fused_topk calls it (srt/layers/moe/topk.py:513-518): softmax over the experts in fp32, the `topk`
    largest (descending; ties -> lower expert id), written INTO topk_weights f32 [T, topk] / topk_ids int32 [T, topk]; `renormalize`
    divides by the sum of the chosen weights.  Torch statement: fused_topk_torch_native (:463-495)."""
    _topk_gate(gating_output, None, topk_weights, topk_ids, 0, renormalize, 1.0, "topk_softmax")


def topk_sigmoid(topk_weights, topk_ids, gating_output, renormalize=False, correction_bias=None):
    """eps.utils.ops._ops.topk_sigmoid: the name srt/layers/moe/topk.py:44-47 resolves at import time (USE_EPS_TOPK_SIGMOID, default on) and
    nothing under python/sglang calls.  The eps sources are not vendored (parity unpinned): the signature mirrors topk_softmax's —
    sigmoid scores, optional selection bias, top-k, optional renormalisation."""
    _topk_gate(gating_output, correction_bias, topk_weights, topk_ids, 1, renormalize, 1.0, "topk_sigmoid")


def routing_flash(router_logits, correction_bias, topk_ids, topk_weights, num_real_experts, routed_scaling_factor=None, renormalize=False):
    """flashinfer.routing_flash as select_experts calls it for LongCat-Flash (srt/layers/moe/topk.py:836-845): the fused form of
    fused_topk_bias (:51-70) — softmax scores, selection by score + correction_bias, weights = the unbiased scores (x routed_scaling_factor).
    `num_real_experts` marks the zero-computation experts (ids >= it): they are selected and weighted like any other, the MoE layer treats
    them (models/longcat_flash.py).  The flashinfer fork's kernel is not vendored: parity is pinned on the torch statement."""
    if correction_bias is not None and correction_bias.numel() != router_logits.shape[1]:
        raise RuntimeError("routing_flash: correction_bias must have one entry per expert (zero experts included)")
    _topk_gate(router_logits, correction_bias, topk_weights, topk_ids, 0, renormalize,
               1.0 if routed_scaling_factor is None else float(routed_scaling_factor), "routing_flash")
