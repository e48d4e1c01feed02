"""BASELINE config 5 beyond attention: ONE speculative-decode step of a LongCat-Flash-shaped decoder as the reference captures it in a
single graph (speculative/spec_decoding_cuda_graph_runner.py:439-470, eagle_worker.py:154-190) — the TARGET VERIFY pass of a
shortcut-connected MoE layer (models/longcat_flash.py:388-476, 502-585) followed by the draft model's decode steps — data-connected, on
two streams, world 1 (all heads / experts on one GPU; the collectives degenerate to their local kernels):

  verify (s_q = 4 draft tokens per request, T = 4 bs token rows, cache_seqlens = seq + 4; flashmla_backend.py:105-176):
    input_layernorm[0] -> self_attn[0] -> (RS +) post_attention_layernorm[0]                                       main stream
      MoE (router -> EP dispatch -> quant -> grouped w13 -> SiLU*mul -> quant -> grouped w2 -> combine)           main stream   ||
      mlps[0] -> input_layernorm[1] -> self_attn[1] -> post_attention_layernorm[1] -> mlps[1]                     SECOND stream
    join: C6 pattern kRSAddResidualRMSNorm with add_in = the MoE output (longcat_flash.py:470-474) = the next layer's input norm
  draft (3 steps, s_q = 1, cache_seqlens = seq + 4 + i + 1; FlashMLAMultiStepDecodeBackend, flashmla_backend.py:411-477):
    input norm -> attention -> post norm -> dense MLP                    (the MTP layer is a dense layer: nextn_use_scmoe = False)

`self_attn` = the absorbed MLA block of cfg4_layer.py: fused_qkv_a (Q2 + G4) -> C7 dual RMSNorm (+ quant) -> q_b (G4) -> A2 (bmm + RoPE +
K5 + K4) -> K1 -> bmm(attn, w_vc) -> o_proj; the dense MLPs thread fp8 scales through the quant-linear API as longcat_flash.py:124-137:
gate_up_proj(x, scale) -> silu_and_mul_fuse_block_quant -> down_proj(x, scale).
Dimensions: hidden 6144, 64 heads, q_lora 1536 / kv_lora 512 / rope 64, dense inter 12288, 512 routed experts x inter 2048, top-12
(the router here is the path's R1 kernel with one expert group; LongCat's zero-computation experts are routing-side and not modelled).
`build()` returns the step, per-stage callables for the checker, and the state."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HID, HEADS, INTER_D, INTER_E, E, TOPK = 6144, 64, 12288, 2048, 512, 12
Q_LORA, KV_LORA, D_ROPE, D_NOPE, D_V = 1536, 512, 64, 128, 128
QKV_A = Q_LORA + KV_LORA + D_ROPE
DRAFT, STEPS = 4, 3


def build(dev, bs=64, seq=16384, seed=0, realistic=False, experts=E):
    import bench
    import cfg4_layer as C4
    import deep_gemm
    import flash_mla_fp8 as fm
    import flashinfer
    import flashinfer.comm as comm
    from eps.executor import silu
    from eps.fast_ep import AllToAll
    from fluent_mi355.bmm import bmm
    from fluent_mi355.gemm import per_token_group_quant_fp8, silu_and_mul_fuse_block_quant

    h = HEADS
    g = torch.Generator(device=dev).manual_seed(seed)
    Tv = bs * DRAFT

    def fp8w(*shape, fan_in=None):
        *lead, N, K = shape
        if not realistic:
            b = torch.randint(0, 255, tuple(shape), device=dev, generator=g, dtype=torch.int16)
            q = torch.where((b & 0x7F) == 0x7F, b - 1, b).to(torch.uint8).view(torch.float8_e4m3fn)
            return q, torch.rand(*lead, (N + 127) // 128, K // 128, device=dev, generator=g) * 1e-2
        one = torch.randn(N, K, device=dev, generator=g) * (fan_in or K) ** -0.5
        q1, s1 = C4.block_quant_weight(one)
        if not lead:
            return q1, s1
        n = lead[0]
        q = torch.empty(n, N, K, dtype=torch.float8_e4m3fn, device=dev)
        s = torch.empty(n, (N + 127) // 128, K // 128, device=dev)
        for e in range(n):
            r = (e * 5) % (N // 128)
            q[e].view(torch.uint8).copy_(torch.roll(q1.view(torch.uint8), 128 * r, 0))
            s[e].copy_(torch.roll(s1, r, 0))
        return q, s

    # three attention sub-layers with their own caches: target attn[0], attn[1], the draft layer's (one spare page per request for the new tokens)
    wl = bench.build_workload(dev, 3, bs, seq + 64, h, seed=seed + 3)
    pages, bt = wl["pages"], wl["block_table"]
    L0 = torch.full((bs,), seq, dtype=torch.int32, device=dev)

    def locs(first, count):
        t = (first + torch.arange(count, device=dev)).view(1, -1).expand(bs, -1)
        return (bt.gather(1, (t // 64).long()).long() * 64 + t % 64).reshape(-1).to(torch.int32)

    seq_v = L0 + DRAFT
    meta_v, ns_v = fm.get_mla_metadata(seq_v, DRAFT * h, 1)
    loc_v = locs(seq, DRAFT)
    pos_v = (seq + torch.arange(DRAFT, device=dev)).view(1, -1).expand(bs, -1).reshape(-1).to(torch.int64)
    seq_d = [L0 + DRAFT + i + 1 for i in range(STEPS)]
    md = [fm.get_mla_metadata(seq_d[i], h, 1) for i in range(STEPS)]
    loc_d = [locs(seq + DRAFT + i, 1) for i in range(STEPS)]
    pos_d = [torch.full((bs,), seq + DRAFT + i, dtype=torch.int64, device=dev) for i in range(STEPS)]
    cache = C4.cos_sin_cache(dev, max_pos=seq + 128)
    _, ws_one = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(0, 1, Tv, HID)

    def attn_weights():
        wkc = (torch.randn(h, KV_LORA, D_NOPE, device=dev, generator=g) * D_NOPE ** -0.5).to(torch.bfloat16)
        wvc = (torch.randn(h, D_V, KV_LORA, device=dev, generator=g) * KV_LORA ** -0.5).to(torch.bfloat16)
        return dict(qkv_a=fp8w(QKV_A, HID), q_b=fp8w(h * (D_NOPE + D_ROPE), Q_LORA), o=fp8w(HID, h * D_V),
                    w_kc=wkc.transpose(1, 2), w_vc=wvc.transpose(1, 2),
                    gamma_q=(1 + 0.1 * torch.randn(Q_LORA, device=dev, generator=g)).to(torch.bfloat16),
                    gamma_kv=(1 + 0.1 * torch.randn(KV_LORA, device=dev, generator=g)).to(torch.bfloat16))

    def mlp_weights():
        return dict(gu=fp8w(2 * INTER_D, HID), down=fp8w(HID, INTER_D))

    ones = lambda: torch.ones(HID, dtype=torch.bfloat16, device=dev)
    W = dict(attn=[attn_weights() for _ in range(3)], mlp=[mlp_weights() for _ in range(3)], gam=[ones() for _ in range(7)],
             router=(torch.randn(experts, HID, device=dev, generator=g) * HID ** -0.5).to(torch.bfloat16), bias=torch.zeros(experts, device=dev),
             w13=fp8w(experts, 2 * INTER_E, HID), w2=fp8w(experts, HID, INTER_E))
    a2a = AllToAll(TOPK, experts, HID, Tv, None)
    rows = a2a.cap * TOPK
    mp = (rows + experts * 31) // 32 * 32
    bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)

    def scales(T, K):
        return torch.empty((K // 128, (T + 3) // 4 * 4), dtype=torch.float32, device=dev).permute(-1, -2)[:T]

    def attn_buffers(T):
        return dict(qkv=bf(T, QKV_A), qa=bf(T, Q_LORA), qa8=torch.empty(T, Q_LORA, dtype=torch.float8_e4m3fn, device=dev), qa8s=scales(T, Q_LORA),
                    q=bf(T, h, D_NOPE + D_ROPE), av=bf(T, h * D_V), out=bf(T, HID))

    def mlp_buffers(T):
        return dict(gu=bf(T, 2 * INTER_D), a8=torch.empty(T, INTER_D, dtype=torch.float8_e4m3fn, device=dev), a8s=scales(T, INTER_D), out=bf(T, HID))

    B = dict(x=torch.randn(Tv, HID, device=dev, generator=g).to(torch.bfloat16), res=torch.randn(Tv, HID, device=dev, generator=g).to(torch.bfloat16),
             n=[bf(Tv, HID) for _ in range(4)], r=[bf(Tv, HID) for _ in range(4)], at=[attn_buffers(Tv), attn_buffers(Tv)],
             ml=[mlp_buffers(Tv), mlp_buffers(Tv)], ex=torch.empty(experts + 1, dtype=torch.int32, device=dev),
             xrows=torch.zeros(rows, HID, dtype=torch.bfloat16, device=dev), xq=torch.empty(rows, HID, dtype=torch.float8_e4m3fn, device=dev),
             xs=torch.empty((HID // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2), gate_up=bf(rows, 2 * INTER_E),
             dq=torch.empty(rows, INTER_E, dtype=torch.float8_e4m3fn, device=dev),
             ds=torch.empty((INTER_E // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2), down=bf(rows, HID), moe=bf(Tv, HID),
             out_n=bf(Tv, HID), out_r=bf(Tv, HID),
             # draft steps
             dx=[torch.randn(bs, HID, device=dev, generator=g).to(torch.bfloat16) for _ in range(STEPS)],
             dres=[torch.randn(bs, HID, device=dev, generator=g).to(torch.bfloat16) for _ in range(STEPS)],
             dn=[[bf(bs, HID), bf(bs, HID)] for _ in range(STEPS)], dr=[[bf(bs, HID), bf(bs, HID)] for _ in range(STEPS)],
             dat=[attn_buffers(bs) for _ in range(STEPS)], dml=[mlp_buffers(bs) for _ in range(STEPS)])
    alt = torch.cuda.Stream(device=dev)

    def norm(x, res_in, gamma, res_out, norm_out, add_in=None):
        """residual add (+ add_in) + RMSNorm: C5 / C6 at world 1 (layernorm.py forward_with_*_fusion)"""
        T = x.shape[0]
        if add_in is None:
            comm.trtllm_allreduce_fusion(allreduce_in=x, world_size=1, world_rank=0, token_num=T, hidden_dim=HID, workspace_ptrs=ws_one,
                                         pattern_code=comm.AllReduceFusionPattern.kARResidualRMSNorm, residual_in=res_in,
                                         residual_out=res_out, norm_out=norm_out, rms_gamma=gamma, rms_eps=1e-6)
        else:
            comm.trtllm_reducescatter_fusion(reducescatter_in=x, world_size=1, world_rank=0, token_num=T, hidden_dim=HID,
                                             workspace_ptrs=ws_one, num_token_current_rank=T,
                                             pattern_code=comm.ReduceScatterFusionPattern.kRSAddResidualRMSNorm, add_in=add_in,
                                             residual_in=res_in, residual_out=res_out, norm_out=norm_out, rms_gamma=gamma, rms_eps=1e-6)

    def attention(x, w, b, layer, s_q, seqlens, meta, ns, positions, loc):
        """the absorbed MLA block on T = bs * s_q token rows -> b["out"] [T, HID]"""
        T = x.shape[0]
        k_lora, k_scale, k_rope = wl["caches"][layer]
        q8, s8 = per_token_group_quant_fp8(x, column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt((q8, s8), w["qkv_a"], b["qkv"])
        comm.trtllm_allgather_fusion(allgather_in=b["qkv"], world_size=1, world_rank=0, hidden_dim=QKV_A, workspace_ptrs=ws_one,
                                     num_token_current_rank=T, allgather_out=b["qkv"], num_token_all_group=T,
                                     pattern_code=comm.AllGatherFusionPattern.kAllGatherfusedRMSFP8BlockWiseQuant, x_norm_out=b["qa"],
                                     quant_out=b["qa8"], scale_out=b["qa8s"], x_rms_gamma=w["gamma_q"], y_rms_gamma=w["gamma_kv"],
                                     x_rms_eps=1e-6, y_rms_eps=1e-6, q_lora_rank=Q_LORA, kv_lora_rank=KV_LORA, qk_rope_head_dim=D_ROPE)
        deep_gemm.gemm_fp8_fp8_bf16_nt((b["qa8"], b["qa8s"]), w["q_b"], b["q"].view(T, h * (D_NOPE + D_ROPE)))
        qn, qs, qr = fm.absorb_rope_quant(b["q"], w["w_kc"], positions, cache, latent_cache=b["qkv"][:, Q_LORA:], k_lora_cache=k_lora,
                                          k_lora_scale_cache=k_scale, k_rope_cache=k_rope, indices=loc, is_neox=False)
        b["qn"], b["qs"], b["qr"] = qn, qs, qr
        o, _ = fm.flash_mla_ckv_fp8_per_token(qn.view(bs, s_q, h, KV_LORA), qr.view(bs, s_q, h, D_ROPE), k_lora.view(pages, 64, 1, KV_LORA),
                                              k_rope.view(pages, 64, 1, D_ROPE), qs.view(bs, s_q, h, 1), k_scale.view(pages, 64, 1, 1),
                                              bt, seqlens, KV_LORA, meta, ns, bench.SCALE, True)
        b["mla_o"] = o
        bmm(o.view(T, h, KV_LORA).transpose(0, 1), w["w_vc"], out=b["av"].view(T, h, D_V).transpose(0, 1))
        oq, os_ = per_token_group_quant_fp8(b["av"], column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt((oq, os_), w["o"], b["out"])
        return b["out"]

    def mlp(x, w, b):
        hq, hs = per_token_group_quant_fp8(x, column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt((hq, hs), w["gu"], b["gu"])
        silu_and_mul_fuse_block_quant(b["gu"], b["a8s"], b["a8"])
        deep_gemm.gemm_fp8_fp8_bf16_nt((b["a8"], b["a8s"]), w["down"], b["out"])
        return b["out"]

    def moe(x):
        logits = flashinfer.dsv3_router_gemm(x, W["router"], out_dtype=torch.float32)
        tw, ti = flashinfer.moe_fused_gate(logits, W["bias"], 1, 1, TOPK, routed_scaling_factor=1.0)
        B["logits"], B["topk_w"], B["topk_ids"] = logits, tw, ti
        a2a.dispatch(out_exclusive_sum=B["ex"], out_expert_x=B["xrows"], dp_x=x, indices=ti, num_global_tokens=Tv, weights=tw)
        flashinfer.quantization.quant_1x128(B["xrows"], B["xq"], B["xs"], B["ex"], experts, (rows + 3) // 4 * 4, mp, HID)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((B["xq"], B["xs"]), W["w13"], B["gate_up"], B["ex"], use_pdl=True)
        act = silu(B["gate_up"], B["ex"], rows)
        flashinfer.quantization.quant_1x128(act, B["dq"], B["ds"], B["ex"], experts, (rows + 3) // 4 * 4, mp, INTER_E)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((B["dq"], B["ds"]), W["w2"], B["down"], B["ex"], use_pdl=True)
        a2a.combine(out_tokens=B["moe"], weights=tw, expert_y=B["down"], num_global_tokens=Tv)
        return B["moe"]

    def dense_branch(x, res):
        """mlps[0] -> input_layernorm[1] -> self_attn[1] -> post_attention_layernorm[1] -> mlps[1] (longcat_flash.py:502-585)"""
        m0 = mlp(x, W["mlp"][0], B["ml"][0])
        norm(m0, res, W["gam"][2], B["r"][2], B["n"][2])
        a1 = attention(B["n"][2], W["attn"][1], B["at"][1], 1, DRAFT, seq_v, meta_v, ns_v, pos_v, loc_v)
        norm(a1, B["r"][2], W["gam"][3], B["r"][3], B["n"][3])
        return mlp(B["n"][3], W["mlp"][1], B["ml"][1]), B["r"][3]

    def verify():
        norm(B["x"], B["res"], W["gam"][0], B["r"][0], B["n"][0])
        a0 = attention(B["n"][0], W["attn"][0], B["at"][0], 0, DRAFT, seq_v, meta_v, ns_v, pos_v, loc_v)
        norm(a0, B["r"][0], W["gam"][1], B["r"][1], B["n"][1])
        cur = torch.cuda.current_stream()
        alt.wait_stream(cur)
        moe(B["n"][1])                                               # main stream
        with torch.cuda.stream(alt):
            d, r = dense_branch(B["n"][1], B["r"][1])                # second stream
        cur.wait_stream(alt)
        # the join: hidden + moe_hidden folded into the next norm (C6 with add_in, longcat_flash.py:470-474)
        norm(d, r, W["gam"][4], B["out_r"], B["out_n"], add_in=B["moe"])
        return B["out_n"]

    def draft(i):
        norm(B["dx"][i], B["dres"][i], W["gam"][5], B["dr"][i][0], B["dn"][i][0])
        a = attention(B["dn"][i][0], W["attn"][2], B["dat"][i], 2, 1, seq_d[i], md[i][0], md[i][1], pos_d[i], loc_d[i])
        norm(a, B["dr"][i][0], W["gam"][6], B["dr"][i][1], B["dn"][i][1])
        return mlp(B["dn"][i][1], W["mlp"][2], B["dml"][i])

    def step():
        verify()
        for i in range(STEPS):
            draft(i)

    info = dict(bs=bs, seq=seq, heads=h, hidden=HID, experts=experts, top_k=TOPK, verify_tokens=Tv, routed_row_bound=rows,
                kv_bytes=2 * bench.algorithmic_bytes(bs, seq + DRAFT, h, DRAFT) + sum(bench.algorithmic_bytes(bs, seq + DRAFT + i + 1, h, 1) for i in range(STEPS)),
                weight_bytes=sum(w[k][0].numel() for w in W["attn"] for k in ("qkv_a", "q_b", "o")) + sum(w[k][0].numel() for w in W["mlp"] for k in ("gu", "down"))
                + W["w13"][0].numel() + W["w2"][0].numel())
    state = dict(W=W, B=B, wl=wl, seq_v=seq_v, loc_v=loc_v, pos_v=pos_v, cache=cache, res_in=B["res"].clone(), pages=pages)
    return step, dict(verify=verify, draft=draft, moe=moe, dense_branch=dense_branch, mlp=mlp, attention=attention, norm=norm), info, state


def main():
    import json
    dev = torch.device("cuda:0")
    bs, seq = int(os.environ.get("BS", 64)), int(os.environ.get("SEQ", 16384))
    step, parts, info, st = build(dev, bs, seq)
    step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return round(a.elapsed_time(b) / n, 4)

    B, W = st["B"], st["W"]
    stage = {"verify (one ScMoE layer, two streams, eager)": timed(parts["verify"]),
             "  MoE branch alone": timed(lambda: parts["moe"](B["n"][1])),
             "  dense branch alone (mlp -> norm -> attn[1] -> norm -> mlp)": timed(lambda: parts["dense_branch"](B["n"][1], B["r"][1])),
             "draft step (norm, attention, norm, dense MLP)": timed(lambda: parts["draft"](0))}
    print(json.dumps({"workload": f"BASELINE config 5: LongCat-shaped spec-decode step, bs={bs} seq={seq}, verify s_q={DRAFT} of one shortcut-connected MoE layer "
                                  f"(MoE || dense branch on two streams, joined by C6 add_in) + {STEPS} draft decode steps, ONE hipGraph, world 1",
                      "ms_per_step_graph": round(ms, 4), "stage_ms_eager": stage, **info}))


if __name__ == "__main__":
    main()
