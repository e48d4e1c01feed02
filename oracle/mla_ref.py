"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (torch-CPU / numpy) of the reference's MLA-decode hot path, function by
function, each citing the reference file:line it follows.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module, and
only as the checker / reported CPU baseline.

Pinning status:
  * quantize_and_cache_k / dequantize_ckv / kv_slot / torch_native_decode are PINNED against
    outputs of the reference itself run in the build container (tests/golden/*.npz made by
    oracle/gen_golden.py, which imports /root/reference/python/sglang).
  * The attention arithmetic of `flash_mla_fp8` / `flash_mla_swap` lives in un-vendored
    submodules (meituan-longcat/FlashMLA @ feature/ckv_fp8_per_token, feature/swapAB; empty
    directories in the reference tree, no SHA) — "parity unpinned" by any reference test.  The
    functional oracle is exact softmax attention over the dequantised cache (what the
    reference's TorchNativeAttnBackend computes), pinned through torch_native_decode.
  * quantize_q_per_token_head has NO in-tree statement (flashmla_backend.py:206 calls the
    absent module).  It is restated by symmetry with the K-side fallback
    (memory_pool.py:873-880): "parity unpinned"; only the (quantise -> attend) composition is
    checked, against exact attention.
"""
from __future__ import annotations

import math

import numpy as np
import torch

PAGE_SIZE = 64  # reference: layers/attention/flashmla_backend.py:25
FP8_MAX = 448.0  # OCP e4m3fn (gfx950 and the reference's NVIDIA data agree)


# --------------------------------------------------------------------------------------
# Page-table arithmetic — bit-exact (reference: mem_cache/allocator.py:60-102, a9)
# --------------------------------------------------------------------------------------
def kv_slot(req_to_page: torch.Tensor, req: int, t: int, page_size: int = PAGE_SIZE) -> int:
    """Token t of request `req` lives at slot req_to_page[req, t//page]*page + t%page."""
    return int(req_to_page[req, t // page_size]) * page_size + t % page_size


def alloc_kv_loc(req_to_page_row: torch.Tensor, new_pages, need_size: int, alloced_len: int,
                 page_size: int = PAGE_SIZE):
    """Restates KVAllocator.alloc (allocator.py:60-102) for one request: returns kv_loc
    (int32 slots for tokens [alloced_len, alloced_len+need_size)) and writes the new page
    ids into the block-table row.  `new_pages`: iterable of free page ids in allocation order."""
    page_offset = alloced_len % page_size
    page_num = (alloced_len + page_size - 1) // page_size
    last_page_remain = page_num * page_size - alloced_len
    last_page_id = int(req_to_page_row[page_num - 1])  # page_num==0 -> index -1 like the reference
    kv_loc = last_page_id * page_size + page_offset + torch.arange(
        0, min(last_page_remain, need_size), dtype=torch.int32)
    if last_page_remain >= need_size:
        return kv_loc
    remain = need_size - last_page_remain
    n_new = (remain + page_size - 1) // page_size
    new_pages = torch.as_tensor(list(new_pages)[:n_new], dtype=torch.int32)
    req_to_page_row[page_num:page_num + n_new] = new_pages
    kv_loc1 = (new_pages.unsqueeze(1) * page_size + torch.arange(0, page_size, dtype=torch.int32))
    return torch.concat([kv_loc, kv_loc1.flatten()[:remain]])


# --------------------------------------------------------------------------------------
# K5: per-token FP8 quantise + scatter (reference: mem_cache/memory_pool.py:873-880, a7)
# --------------------------------------------------------------------------------------
def quantize_and_cache_k(key, k_lora_cache, k_lora_scale_cache, k_rope_cache, indices,
                         head_dim_v: int = 512):
    """key [n,1,576] (model dtype); caches as in MLATokenToKVPool per_token_head
    (memory_pool.py:635-649): k_lora u8/fp8 [S,1,512], scale f32 [S,1,1], rope [S,1,64]."""
    k_lora = key[..., :head_dim_v].float()
    k_rope = key[..., head_dim_v:].float()
    scale = k_lora.abs().amax(dim=-1, keepdim=True).clamp(1e-26) / FP8_MAX
    q = (k_lora / scale).to(torch.float8_e4m3fn)
    r = (k_rope / scale).to(k_rope_cache.dtype)
    idx = indices.long()
    k_lora_cache.view(torch.uint8)[idx] = q.view(torch.uint8)
    k_lora_scale_cache[idx] = scale
    k_rope_cache[idx] = r


# --------------------------------------------------------------------------------------
# K6: gather + dequantise (reference: mem_cache/memory_pool.py:826-831, a8)
# --------------------------------------------------------------------------------------
def dequantize_ckv_fused_indexed(k_lora_fp8, k_rope, k_scale, indices, out_dtype=torch.bfloat16):
    idx = indices.long()
    k_lora = k_lora_fp8.view(torch.float8_e4m3fn)[idx].float()
    scale = k_scale[idx]
    rope = k_rope[idx].float()
    return (k_lora * scale).to(out_dtype).contiguous(), (rope * scale).to(out_dtype).contiguous()


# --------------------------------------------------------------------------------------
# K4: per-(token, head) FP8 quantisation of Q.  No in-tree statement (call site
# flashmla_backend.py:125,206); restated by symmetry with K5: scale = amax|q_nope|/448 (clamped),
# q_nope -> e4m3fn, q_rope -> rope/scale in the model dtype.  PARITY UNPINNED.
# --------------------------------------------------------------------------------------
def quantize_ckv_per_token_head(q, kv_lora_rank: int = 512):
    """q [bs,s_q,H,576] -> (q_nope fp8 [bs,s_q,H,512], q_scale f32 [bs,s_q,H,1], q_rope [bs,s_q,H,64])."""
    qn = q[..., :kv_lora_rank].float()
    qr = q[..., kv_lora_rank:].float()
    scale = qn.abs().amax(dim=-1, keepdim=True).clamp(1e-26) / FP8_MAX
    return (qn / scale).to(torch.float8_e4m3fn), scale, (qr / scale).to(q.dtype)


# --------------------------------------------------------------------------------------
# Exact paged MLA attention (the functional oracle for K1/K2).  Equivalent to what
# TorchNativeAttnBackend._run_sdpa_forward_decode (torch_native_backend.py:275-343) computes per
# request — gather the request's tokens through the page table, softmax(q·kᵀ·scale)·v — but in
# fp64 so that it is an accuracy reference, and with FlashMLA's causal rule for s_q > 1
# (flashmla_backend.py:135-136: cache_seqlens already includes the s_q new tokens; query j sees
# keys [0, seqlen - (s_q-1-j)) ).
# --------------------------------------------------------------------------------------
def mla_decode_exact(q, k_full, block_table, cache_seqlens, softmax_scale, head_dim_v=512,
                     causal=True, page_size=PAGE_SIZE):
    """q float [bs,s_q,H,D]; k_full float [num_slots, D] (already dequantised);
    returns o f64 [bs,s_q,H,dv], lse f64 [bs,H,s_q] (natural log; -inf for empty rows)."""
    bs, s_q, H, D = q.shape
    o = torch.zeros(bs, s_q, H, head_dim_v, dtype=torch.float64)
    lse = torch.full((bs, H, s_q), -math.inf, dtype=torch.float64)
    kf = k_full.double()
    for b in range(bs):
        L = int(cache_seqlens[b])
        if L <= 0:
            continue
        t = torch.arange(L)
        slots = block_table[b, (t // page_size).long()].long() * page_size + (t % page_size)
        kb = kf[slots]  # [L, D]
        for j in range(s_q):
            Lj = L - (s_q - 1 - j) if causal else L
            if Lj <= 0:
                continue
            s = (q[b, j].double() @ kb[:Lj].T) * softmax_scale  # [H, Lj]
            m = s.max(dim=-1, keepdim=True).values
            p = torch.exp(s - m)
            l = p.sum(-1, keepdim=True)
            o[b, j] = (p / l) @ kb[:Lj, :head_dim_v]
            lse[b, :, j] = (m + torch.log(l)).squeeze(-1)
    return o, lse


def mla_decode_fp8_per_token(q_nope, q_scale, q_rope, k_lora, k_scale, k_rope, block_table,
                             cache_seqlens, softmax_scale, causal=True):
    """Oracle for K1 (flash_mla_ckv_fp8_per_token, call site flashmla_backend.py:208-222):
    dequantise exactly (fp64) then exact attention.  k_lora [pages,64,1,512] fp8/u8,
    k_scale [pages,64,1,1], k_rope [pages,64,1,64]; q_* from quantize_ckv_per_token_head."""
    qs = q_scale.double()
    qn = q_nope.view(torch.float8_e4m3fn).double() * qs
    qr = q_rope.double() * qs
    q = torch.cat([qn, qr], dim=-1)
    ks = k_scale.reshape(-1, 1).double()
    kn = k_lora.view(torch.float8_e4m3fn).reshape(-1, k_lora.shape[-1]).double() * ks
    kr = k_rope.reshape(-1, k_rope.shape[-1]).double() * ks
    k = torch.cat([kn, kr], dim=-1)
    return mla_decode_exact(q, k, block_table, cache_seqlens, softmax_scale, k_lora.shape[-1], causal)


def mla_decode_with_kvcache(q, k_cache, block_table, cache_seqlens, head_dim_v, softmax_scale,
                            causal=True, descale_q=None, descale_k=None):
    """Oracle for K2 (flash_mla_with_kvcache, call sites flashmla_backend.py:227-254):
    q [bs,s_q,H,576] bf16 or fp8; k_cache [pages,64,1,576] bf16 or fp8(u8)."""
    dq = float(descale_q) if descale_q is not None else 1.0
    dk = float(descale_k) if descale_k is not None else 1.0
    qf = (q.view(torch.float8_e4m3fn) if q.dtype == torch.uint8 else q).double() * dq
    kc = k_cache.view(torch.float8_e4m3fn) if k_cache.dtype == torch.uint8 else k_cache
    kf = kc.reshape(-1, kc.shape[-1]).double() * dk
    return mla_decode_exact(qf, kf, block_table, cache_seqlens, softmax_scale, head_dim_v, causal)


# --------------------------------------------------------------------------------------
# The reference's CPU path itself (CPU baseline "port"): per-request gather + SDPA loop,
# restating TorchNativeAttnBackend._run_sdpa_forward_decode (torch_native_backend.py:309-343)
# as called from forward_decode (:472-528) for the MLA layer (k == v buffer, v = first 512 dims;
# memory_pool.py:841-852).
# --------------------------------------------------------------------------------------
def torch_native_decode(q, kv_buffer, req_to_token, req_pool_indices, seq_lens, scaling,
                        v_head_dim=512):
    """q [bs, H, 576] (model dtype); kv_buffer [slots, 1, 576]; returns o [bs, H*512]."""
    from torch.nn.functional import scaled_dot_product_attention as sdpa

    bs, H, D = q.shape
    o = q.new_empty((bs, H, v_head_dim))
    k_cache = kv_buffer
    v_cache = kv_buffer[..., :v_head_dim]
    query = q.movedim(0, q.dim() - 2)  # [H, bs, D]
    start_q = 0
    for seq_idx in range(seq_lens.shape[0]):
        seq_len_kv = int(seq_lens[seq_idx])
        end_q = start_q + 1
        per_req_query = query[:, start_q:end_q, :]
        per_req_tokens = req_to_token[req_pool_indices[seq_idx], :seq_len_kv].long()
        per_req_key = k_cache[per_req_tokens].movedim(0, query.dim() - 2)
        per_req_value = v_cache[per_req_tokens].movedim(0, query.dim() - 2)
        per_req_out = sdpa(per_req_query.unsqueeze(0), per_req_key.unsqueeze(0),
                           per_req_value.unsqueeze(0), enable_gqa=True, scale=scaling,
                           is_causal=False).squeeze(0).movedim(query.dim() - 2, 0)
        o[start_q:end_q] = per_req_out
        start_q = end_q
    return o.reshape(bs, H * v_head_dim)


# --------------------------------------------------------------------------------------
# K3: split-KV tile scheduler.  The metadata FORMAT is ours (only our kernel consumes it;
# reference call sites flashmla_backend.py:261-265,307-321 only fix the SHAPES to be static in
# bs).  This is the Python statement of csrc/mla_metadata.hip, compared bit-exactly in tests.
# --------------------------------------------------------------------------------------
META_W = 8
FIXED_OVERHEAD_TILES = 2


MIN_SPLIT_CAP, PAGES_PER_SPLIT = 32, 8


def _parts_needed(ntiles, capacity, limit):
    """Parts the greedy walk needs with this capacity (stops counting beyond `limit`)."""
    req, tile, parts, bs = 0, 0, 0, len(ntiles)
    while req < bs and parts <= limit:
        remain = capacity
        parts += 1
        while req < bs:
            left = ntiles[req] - tile
            if remain >= left + FIXED_OVERHEAD_TILES:
                remain -= left + FIXED_OVERHEAD_TILES
                req, tile = req + 1, 0
            else:
                take = remain - FIXED_OVERHEAD_TILES
                if take > 0:
                    tile += take
                break
    return parts


def get_mla_metadata(cache_seqlens, num_parts: int, page_size: int = PAGE_SIZE):
    """Greedy partition of the row-major (request, 64-token tile) list into `num_parts` contiguous
    parts of capacity P = the smallest value in [ceil(total/num_parts), +63] for which the greedy walk
    places everything (fallback: ceil(total/num_parts) + FIXED_OVERHEAD).  A piece of a request costs
    its tiles + FIXED_OVERHEAD.  Returns (meta int32 [num_parts, 8], num_splits int32 [bs+1]).
    meta row = [begin_req, begin_tile, end_req, end_tile, begin_split_idx, 0, 0, 0]: the part
    covers tiles (begin_req, begin_tile) .. (end_req, end_tile) exclusive; begin_req == bs means
    "no work".  num_splits is cumulative: request b owns accumulator slots
    [num_splits[b], num_splits[b+1])."""
    seqlens = [int(x) for x in cache_seqlens]
    bs = len(seqlens)
    ntiles = [((L + page_size - 1) // page_size) if L > 0 else 0 for L in seqlens]
    total = sum(n + FIXED_OVERHEAD_TILES for n in ntiles)
    p_min = max((total + num_parts - 1) // num_parts, 1 + FIXED_OVERHEAD_TILES)
    # no request in more than max(32, pages/8) parts: a part of one or two pages is all prologue / epilogue, and the merge of
    # a row grows with its split count (bs=1, seq=16384 on 256 parts: 157 us per layer before)
    nt_max = max(ntiles, default=0)
    split_cap = max(MIN_SPLIT_CAP, nt_max // PAGES_PER_SPLIT)
    p_min = max(p_min, (nt_max + split_cap - 1) // split_cap + FIXED_OVERHEAD_TILES)
    payload = p_min + FIXED_OVERHEAD_TILES
    for cand in range(p_min, p_min + 64):
        if _parts_needed(ntiles, cand, num_parts) <= num_parts:
            payload = cand
            break
    meta = np.zeros((num_parts, META_W), dtype=np.int32)
    num_splits = np.zeros(bs + 1, dtype=np.int32)
    req, tile, split, cum = 0, 0, 0, 0
    for p in range(num_parts):
        meta[p, 0], meta[p, 1], meta[p, 4] = req, tile, split
        remain = payload
        while req < bs:
            left = ntiles[req] - tile
            if remain >= left + FIXED_OVERHEAD_TILES or p == num_parts - 1:
                remain -= left + FIXED_OVERHEAD_TILES
                cum += split + 1
                num_splits[req + 1] = cum
                req, tile, split = req + 1, 0, 0
            else:
                take = remain - FIXED_OVERHEAD_TILES
                if take > 0:
                    tile += take
                    split += 1
                break
        meta[p, 2], meta[p, 3] = req, tile
    assert req == bs
    return meta, num_splits


def get_mla_metadata_positions(cache_seqlens, num_parts: int, page_size: int = PAGE_SIZE):
    """The SAME partition as get_mla_metadata, stated the way csrc/mla_metadata.hip's one-pass kernel computes it (round 6): on the
    cost axis C[r] = sum_{k<r} (tiles_k + FIXED_OVERHEAD) a part that starts at (req, tile) sits at x = C[req] + tile, ends at y = x + P, and
    the next part starts at x' = max(y - FIXED_OVERHEAD, C[r']) with r' the first request the part does not finish (C[r' + 1] > y).  Every
    candidate capacity walks ONCE and records (r', tile, split) per part; the rows and the split counts follow from the records of the
    winning capacity without a second walk.  tests/test_oracle_golden.py checks it against get_mla_metadata on random batches."""
    seqlens = [int(x) for x in cache_seqlens]
    bs = len(seqlens)
    OH = FIXED_OVERHEAD_TILES
    ntiles = [((L + page_size - 1) // page_size) if L > 0 else 0 for L in seqlens]
    C = [0]
    for n in ntiles:
        C.append(C[-1] + n + OH)
    total = C[-1]
    p_min = max((total + num_parts - 1) // num_parts, 1 + OH)
    nt_max = max(ntiles, default=0)
    split_cap = max(MIN_SPLIT_CAP, nt_max // PAGES_PER_SPLIT)
    p_min = max(p_min, (nt_max + split_cap - 1) // split_cap + OH)

    def walk(P):
        r, p, split, y = 0, 0, 0, P
        rec = [(0, 0, 0)]                         # (req, tile, split) at the start of part p
        while r < bs and p < num_parts:
            if C[r + 1] <= y:                     # the part finishes request r
                r, split = r + 1, 0
            else:                                 # the part closes inside (or in front of) request r
                x = max(y - OH, C[r])
                split += 1 if y - OH > C[r] else 0
                p += 1
                rec.append((r, x - C[r], split))
                y = x + P
        return r == bs, rec[:num_parts]

    rec = None
    for cand in range(p_min, p_min + 64):
        ok, rc = walk(cand)
        if ok:
            rec = rc
            break
    if rec is None:
        rec = walk(p_min + OH)[1]                  # the last part takes whatever is left
    last = len(rec) - 1
    meta = np.zeros((num_parts, META_W), dtype=np.int32)
    touched = [1] * bs
    for p in range(num_parts):
        b = rec[p] if p <= last else (bs, 0, 0)
        e = rec[p + 1] if p + 1 <= last else (bs, 0, 0)
        meta[p, 0], meta[p, 1], meta[p, 4] = b[0], b[1], b[2]
        meta[p, 2], meta[p, 3] = e[0], e[1]
        if p <= last and e[0] > b[0]:
            touched[b[0]] = b[2] + 1
    num_splits = np.zeros(bs + 1, dtype=np.int32)
    num_splits[1:] = np.cumsum(np.asarray(touched, dtype=np.int64)) if bs else []
    return meta, num_splits


# --------------------------------------------------------------------------------------
# Bit-level statement of OUR kernel's arithmetic (csrc/mla_decode_fp8.hip) — NOT of the reference.
# The exact oracle above bounds the end-to-end error (FP8 tolerance); this one pins the design:
# per page, the two 32-token halves keep independent integer references m_W = ceil(running max
# of y)+2, y = s*log2e + log2(k_scale[t]); P' = 2^(y - m_W + 8) is rounded to e4m3 (RNE);
# O accumulates 2^(m_W - M) * P'_q * k8 (exact powers of two).  Two normalisers per half: lq_W sums
# the ROUNDED P'_q/k_scale[t] (divides O: the weights of the quantised numerator sum to exactly 1,
# so e.g. a 1-token sequence returns its latent exactly), lx_W sums the unrounded P'/k_scale[t]
# (exact LSE).  Differences to the GPU are fp32-vs-fp64 accumulation and exp2 ulps only.
# --------------------------------------------------------------------------------------
def mla_decode_fp8_per_token_emulated(q_nope, q_scale, q_rope, k_lora, k_scale, k_rope, block_table,
                                      cache_seqlens, softmax_scale, causal=True, page_size=PAGE_SIZE):
    bs, s_q, H, dn = q_nope.shape
    rows = s_q * H
    q8 = q_nope.view(torch.float8_e4m3fn).double().reshape(bs, rows, dn)
    qr = q_rope.double().reshape(bs, rows, -1)
    qs = (q_scale.float().reshape(bs, rows) * torch.tensor(softmax_scale * 1.4426950408889634, dtype=torch.float32))
    k8 = k_lora.view(torch.float8_e4m3fn).reshape(-1, dn).double()
    kr = k_rope.reshape(-1, k_rope.shape[-1]).double()
    ks = k_scale.reshape(-1).float()
    out = torch.zeros(bs, rows, dn, dtype=torch.float64)
    lse = torch.full((bs, rows), -math.inf, dtype=torch.float64)
    NEG = -16384.0
    for b in range(bs):
        L = int(cache_seqlens[b])
        if L <= 0:
            continue
        j = torch.arange(rows) // H
        L_row = (L - (s_q - 1 - j)) if causal else torch.full((rows,), L)
        O = torch.zeros(rows, dn, dtype=torch.float64)
        M = torch.full((rows,), NEG, dtype=torch.float64)
        mW = [torch.full((rows,), NEG, dtype=torch.float64) for _ in range(2)]
        lW = [torch.zeros(rows, dtype=torch.float64) for _ in range(2)]   # exact (LSE)
        lQ = [torch.zeros(rows, dtype=torch.float64) for _ in range(2)]   # from rounded P (normalises O)
        for pg in range((L + page_size - 1) // page_size):
            page = int(block_table[b, pg])
            Pq, sc = [], []
            for W in range(2):
                t0 = pg * page_size + 32 * W
                slots = page * page_size + 32 * W + torch.arange(32)
                valid_tok = (t0 + torch.arange(32)) < L
                ksw = torch.where(valid_tok, ks[slots], torch.ones(32))
                acc = (q8[b] @ k8[slots].T + qr[b] @ kr[slots].T).float()                 # [rows, 32]
                y = torch.addcmul(torch.log2(ksw)[None, :], acc * qs[b][:, None], ksw[None, :]).double()
                mask = (t0 + torch.arange(32))[None, :] >= L_row[:, None]
                y = torch.where(mask | torch.isnan(y), torch.full_like(y, -math.inf), y)
                tmax = y.max(dim=1).values
                m_new = torch.where(tmax > mW[W], torch.ceil(tmax) + 2.0, mW[W])
                lW[W] = lW[W] * torch.exp2(mW[W] - m_new)
                lQ[W] = lQ[W] * torch.exp2(mW[W] - m_new)
                mW[W] = m_new
                P = torch.exp2(y - m_new[:, None] + 8.0)
                lW[W] = lW[W] + (P / ksw[None, :].double()).sum(1)
                Pq.append(P.float().to(torch.float8_e4m3fn).double())
                lQ[W] = lQ[W] + (Pq[-1] / ksw[None, :].double()).sum(1)
                sc.append(slots)
            M_new = torch.maximum(M, torch.maximum(mW[0], mW[1]))
            O = O * torch.exp2(M - M_new)[:, None]
            M = M_new
            for W in range(2):
                v = k8[sc[W]].clone()
                v[(pg * page_size + 32 * W + torch.arange(32)) >= L] = 0.0   # zero-filled tail rows
                O = O + torch.exp2(torch.clamp(mW[W] - M, min=-127.0))[:, None] * (Pq[W] @ v)
        l = lW[0] * torch.exp2(mW[0] - M) + lW[1] * torch.exp2(mW[1] - M)
        lq = lQ[0] * torch.exp2(mW[0] - M) + lQ[1] * torch.exp2(mW[1] - M)
        ok = l > 0
        okq = lq > 0
        out[b][okq] = O[okq] / lq[okq][:, None]
        lse[b][ok] = (torch.log2(l[ok]) + M[ok] - 8.0) * math.log(2.0)
    return out.reshape(bs, s_q, H, dn), lse.reshape(bs, s_q, H).permute(0, 2, 1).contiguous()


def move_kv_cache(buffers, tgt_loc, src_loc):
    """`buf[tgt] = buf[src]` for every buffer (memory_pool.py:746-763 for the per_token_head MLA pool,
    move_kv_cache_native :2039-2052): advanced indexing reads every source row before it writes any target row.
    Pinned against tests/golden/kv_move.npz."""
    if tgt_loc.numel() == 0:
        return
    t, s = tgt_loc.view(-1).long(), src_loc.view(-1).long()
    for buf in buffers:
        buf[t] = buf[s]
