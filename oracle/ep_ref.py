"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the DeepExecutor's ep_scatter / ep_gather
(python/sglang/srt/layers/moe/executors/deep_ep_executor.py:173-430, Triton in the reference).  Pinned against
tests/golden/ep_scatter_gather.npz, produced by the reference's own Triton kernels under Triton's CPU interpreter
(oracle/gen_golden.py:gen_ep_scatter_gather).  The reference hands out positions inside an expert's group with atomics
(:247): the order inside a group is unspecified there; this restatement uses program order (token, then k), which is what
the interpreter produces.  Only tests/ may import this."""
import numpy as np


def ep_scatter(recv_x, recv_x_scale, recv_topk, num_recv_tokens_per_expert):
    """-> (expert_start_loc AFTER the call [E], output_tensor [M, H], output_tensor_scale [M, H/128], m_indices [M],
    output_index [T, K]); M = sum(num_recv_tokens_per_expert) (each a multiple of 128, :281)."""
    T, K = recv_topk.shape
    cnt = np.asarray(num_recv_tokens_per_expert, np.int64)
    E, M = cnt.shape[0], int(cnt.sum())
    start = np.cumsum(cnt) - cnt                                         # :192-193
    m_indices = np.full(M, -1, np.int32)
    for e in range(E):
        m_indices[start[e]: start[e] + (cnt[e] + 127) // 128 * 128] = e   # :200-204
    cursor = start.copy()
    out = np.zeros((M, recv_x.shape[1]), recv_x.dtype)
    outs = np.zeros((M, recv_x_scale.shape[1]), recv_x_scale.dtype)
    oidx = np.full((T, K), -1, np.int32)
    for t in range(T):                                                   # :238-257
        for k in range(K):
            e = int(recv_topk[t, k])
            if e >= 0:
                d = int(cursor[e]); cursor[e] += 1                       # atomic_add(expert_start_loc + e, 1)
                oidx[t, k] = d
                out[d], outs[d] = recv_x[t], recv_x_scale[t]
    return cursor.astype(np.int32), out, outs, m_indices, oidx


def ep_gather(input_f32, recv_topk_ids, recv_topk_weight, input_index):
    """out[t] = sum_k (id >= 0) w[t,k] * input[index[t,k]] in fp32, k ascending (:358-381) -> float32 [T, H]
    (the caller rounds to the output dtype)."""
    T, K = recv_topk_ids.shape
    out = np.zeros((T, input_f32.shape[1]), np.float32)
    for t in range(T):
        for k in range(K):
            if recv_topk_ids[t, k] >= 0:
                out[t] += input_f32[int(input_index[t, k])] * np.float32(recv_topk_weight[t, k])
    return out
