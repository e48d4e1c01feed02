"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, float32) of the reference's DeepSeek-V3 router selection,
`biased_grouped_topk_impl` (python/sglang/srt/layers/moe/topk.py:596-663), the torch statement behind the
`moe_fused_gate` kernel its GPU path calls (topk.py:709-733).  Pinned against tests/golden/router_biased_grouped_topk.npz
(generated from the reference's own source by oracle/gen_golden.py:gen_router).  Only tests/ may import this."""
import numpy as np


def biased_grouped_topk(logits, bias, num_expert_group, topk_group, topk, routed_scaling_factor=1.0,
                        apply_routed_scaling_factor_on_output=False, num_token_non_padded=None, renormalize=True):
    """-> (topk_weights f32 [T, topk], topk_ids i32 [T, topk]); rows sorted by descending choice score (the reference's
    torch.topk(sorted=False) leaves the order unspecified — compare as sets)."""
    x = np.asarray(logits, np.float32)
    T, E = x.shape
    scores = (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)            # topk.py:613 gating_output.sigmoid()
    choice = scores + np.asarray(bias, np.float32)[None, :]                              # :616
    grp = choice.reshape(T, num_expert_group, -1)
    top2 = np.sort(grp, axis=-1)[..., -2:] if grp.shape[-1] >= 2 else grp
    group_scores = top2.sum(-1, dtype=np.float32)                                        # :617-621 (top-2 sum per group)
    gidx = np.argsort(-group_scores, axis=-1, kind="stable")[:, :topk_group]             # :622-624
    gmask = np.zeros((T, num_expert_group), bool)
    np.put_along_axis(gmask, gidx, True, axis=1)                                         # :625-626
    emask = np.repeat(gmask, E // num_expert_group, axis=1)                              # :627-631
    masked = np.where(emask, choice, -np.inf).astype(np.float32)                         # :632-634
    ids = np.argsort(-masked, axis=-1, kind="stable")[:, :topk]                          # :636-641
    w = np.take_along_axis(scores, ids, axis=1)                                          # :642 (UNBIASED scores)
    if renormalize:
        w = w / w.sum(-1, keepdims=True, dtype=np.float32)                               # :654-660
        if apply_routed_scaling_factor_on_output:
            w = w * np.float32(routed_scaling_factor)                                    # :661-662
    ids = ids.astype(np.int32)
    if num_token_non_padded is not None:
        ids[np.arange(T) >= int(num_token_non_padded), :] = -1                           # :673-680
    return w.astype(np.float32), ids


def topk_plain(logits, topk, renormalize, bias=None, sigmoid=False, scale=1.0):
    """The plain (ungrouped) routers of the same file: fused_topk_native (topk.py:73-91: softmax over the experts, torch.topk, optional
    renormalisation — the statement behind flashinfer.topk_softmax) and fused_topk_bias (:51-70: selection by softmax + correction_bias,
    weights = the unbiased scores — the statement behind flashinfer.routing_flash).  -> (weights f32 [T, topk], ids i32 [T, topk]), rows by
    descending choice score (ties -> lower id).  sigmoid=True: eps' topk_sigmoid (unpinned, see fluent_mi355/router.py)."""
    x = np.asarray(logits, np.float32).astype(np.float64)
    if sigmoid:
        scores = (1.0 / (1.0 + np.exp(-x))).astype(np.float32)
    else:
        e = np.exp(x - x.max(-1, keepdims=True))
        scores = (e / e.sum(-1, keepdims=True)).astype(np.float32)                        # gating_output.softmax(dim=-1)
    choice = scores if bias is None else scores + np.asarray(bias, np.float32)[None, :]   # :61
    ids = np.argsort(-choice, axis=-1, kind="stable")[:, :topk]                           # :62 / :88
    w = np.take_along_axis(scores, ids, axis=1)                                           # :63 (UNBIASED scores)
    if renormalize:
        w = w / w.sum(-1, keepdims=True, dtype=np.float32)                                # :65-66 / :89-90
    return (w * np.float32(scale)).astype(np.float32), ids.astype(np.int32)
