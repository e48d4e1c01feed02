"""BASELINE config 4 as a multi-rank step (bench.py --mode cfg4): ONE data-connected DeepSeek-V3 decoder layer — attention-TP + EP MoE
with the path's real collectives, one process per GPU — in the order models/deepseek_v2.py runs it (:775-889 attention, :313-346 MoE):

  hidden slice [bs/world, 7168] (reduce-scattered between layers, decoder_comm_manager.py:152-219 RSAG)
  C5   residual add + input RMSNorm on the slice (no exchange)
  Q2   1x128 quant + fused_qkv_a_proj (dense fp8 G4) on the slice                         -> [bs/world, 2112]  (q_a 1536 | kv_a 512 | k_pe 64)
  C7   all-gather + q_a / kv_a RMSNorm (+ 1x128 quant of q_a) in one launch               -> [bs, 2112], q_a fp8 [bs, 1536]
  G4   q_b_proj (column-parallel: this rank's 128/world heads)                            -> q [bs, h, 192]
  A2   bmm(q_nope, w_kc) + RoPE(q_pe, k_pe) + K5 (set_kv_buffer) + K4 (quantise q) in one launch  (FLUENT_CFG4_A2=0: the four-launch chain
       B2 bmm, R2 rope, K5, K4 of the unmodified model code)
  K1   paged FP8 MLA decode over the rank's heads, all bs requests (latent KV replicated under attention-TP)
  B2   bmm(attn, w_vc)                                                                    -> [bs, h * 128]
  Q2 + G4  o_proj (row-parallel: partial sums over the TP group)                          -> [bs, 7168]
  C6   reduce-scatter + residual + post-attention RMSNorm                                 -> [bs/world, 7168]
  MoE  router GEMM (B3) + R1 moe_fused_gate -> EP dispatch (eps.fast_ep.AllToAll) -> quant_1x128 -> grouped w13 (G1) -> SiLU*mul ->
       quant_1x128 -> grouped w2 -> EP combine;  the shared expert (Q2 + G4 gate_up -> silu_and_mul_fuse_block_quant -> G4 down) runs on
       a SECOND stream beside it (deepseek_v2.py:337-340) and is added to the routed output.
Nothing is synthetic between the stages: the query K1 attends with is the projection of the layer's hidden state, o_proj reads K1's output.
Weights are random (`realistic=True`: N(0, 1/fan_in) through the 128 x 128 block quantiser, so that activations stay O(1) and the oracle
composition of tests/test_cfg4_gpu.py is well conditioned; False: random fp8 bytes, timing only).  Static shapes: one hipGraph per step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))

BS, SEQ, HEADS, HID, INTER, E, TOPK = 256, 8192, 128, 7168, 2048, 256, 8
Q_LORA, KV_LORA, D_ROPE, D_NOPE, D_V = 1536, 512, 64, 128, 128
QKV_A = Q_LORA + KV_LORA + D_ROPE   # 2112 (16.5 weight-scale blocks: the last block is half used)
N_GROUP, TOPK_GROUP = 8, 4
MAX_POS = 16384
USE_A2 = os.environ.get("FLUENT_CFG4_A2", "1") != "0"


def cos_sin_cache(dev, max_pos=MAX_POS, dim=D_ROPE, base=10000.0):
    """[max_pos, dim] f32 (cos | sin halves) as rotary_embedding.py:104-115 builds it (plain RoPE frequencies: the scaling variant only
    changes the table's values, which the kernels read as data)."""
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    f = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
    return torch.cat([f.cos(), f.sin()], dim=-1).to(dev).contiguous()


def block_quant_weight(w):
    """f32 [.., N, K] -> (fp8 [.., N, K], f32 scales [.., ceil(N/128), K/128]): the 128 x 128 block format of an FP8 checkpoint."""
    *lead, N, K = w.shape
    Np = (N + 127) // 128 * 128
    wp = torch.zeros(*lead, Np, K, device=w.device)
    wp[..., :N, :] = w
    blk = wp.view(*lead, Np // 128, 128, K // 128, 128)
    amax = blk.abs().amax(dim=(-3, -1), keepdim=True).clamp_min(1e-12)
    q = (blk / (amax / 448.0)).clamp(-448, 448).to(torch.float8_e4m3fn).view(*lead, Np, K)[..., :N, :].contiguous()
    return q, (amax / 448.0).reshape(*lead, Np // 128, K // 128).contiguous()


def build(dev, world, rank, group, layers, seq=SEQ, bs=BS, seed=0, realistic=False):
    """-> (step, info): `step()` runs `layers` decoder layers of this rank's share; `info` names the static sizes."""
    import bench
    import deep_gemm
    import flash_mla_fp8 as fm
    import flashinfer
    import flashinfer.comm as comm
    from eps.executor import silu
    from eps.fast_ep import AllToAll
    from fluent_mi355.bmm import bmm
    from fluent_mi355.gemm import per_token_group_quant_fp8, silu_and_mul_fuse_block_quant
    from fluent_mi355.rope import apply_rope_with_cos_sin_cache_inplace

    if HEADS % world or E % world or bs % world:
        raise SystemExit(f"cfg4: world {world} must divide heads {HEADS}, experts {E} and bs {bs}")
    h, el, t_loc = HEADS // world, E // world, bs // world
    g = torch.Generator(device=dev).manual_seed(seed + 17 * rank)
    gs = torch.Generator(device=dev).manual_seed(seed + 1000)   # replicated tensors (same on every rank)

    def fp8w(shape, gen, fan_in=None):
        """-> (fp8 weights, f32 block scales) of shape [.., N, K]"""
        *lead, N, K = shape
        if not realistic:
            b = torch.randint(0, 255, tuple(shape), device=dev, generator=gen, dtype=torch.int16)
            q = torch.where((b & 0x7F) == 0x7F, b - 1, b).to(torch.uint8).view(torch.float8_e4m3fn)
            return q, torch.rand(*lead, (N + 127) // 128, K // 128, device=dev, generator=gen) * 1e-2
        one = torch.randn(N, K, device=dev, generator=gen) * (fan_in or K) ** -0.5
        q1, s1 = block_quant_weight(one)
        if not lead:
            return q1, s1
        # the other groups are row-rotated copies (distinct memory, same statistics: 256 experts of fresh normals would dominate set-up)
        n = lead[0]
        q = torch.empty(n, N, K, dtype=torch.float8_e4m3fn, device=dev)
        s = torch.empty(n, (N + 127) // 128, K // 128, device=dev)
        for e in range(n):
            r = (e * 5) % (N // 128)
            q[e].view(torch.uint8).copy_(torch.roll(q1.view(torch.uint8), 128 * r, 0))
            s[e].copy_(torch.roll(s1, r, 0))
        return q, s

    wl = bench.build_workload(dev, layers, bs, seq, h, seed=seed + 3)      # same requests / caches on every rank (TP): same seed
    meta, ns = fm.get_mla_metadata(wl["seqlens"], h, 1)
    positions = (wl["seqlens"] - 1).to(torch.int64)
    cache = cos_sin_cache(dev)
    h_tp, ws_tp = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(rank, world, bs, HID, group=group)
    _, ws_one = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(0, 1, bs, HID)        # the exchange-free norm
    W = []
    for _ in range(layers):
        wkc = (torch.randn(h, KV_LORA, D_NOPE, device=dev, generator=g) * D_NOPE ** -0.5).to(torch.bfloat16)   # k-contiguous [H, 512, 128]
        wvc = (torch.randn(h, D_V, KV_LORA, device=dev, generator=g) * KV_LORA ** -0.5).to(torch.bfloat16)     # k-contiguous [H, 128, 512]
        W.append(dict(gamma1=torch.ones(HID, dtype=torch.bfloat16, device=dev), gamma2=torch.ones(HID, dtype=torch.bfloat16, device=dev),
                      gamma_q=(1 + 0.1 * torch.randn(Q_LORA, device=dev, generator=gs)).to(torch.bfloat16),
                      gamma_kv=(1 + 0.1 * torch.randn(KV_LORA, device=dev, generator=gs)).to(torch.bfloat16),
                      qkv_a=fp8w((QKV_A, HID), gs), q_b=fp8w((h * (D_NOPE + D_ROPE), Q_LORA), g),
                      w_kc=wkc.transpose(1, 2), w_vc=wvc.transpose(1, 2),          # [H, 128, 512] / [H, 512, 128] views, as the model holds them
                      o=fp8w((HID, h * D_V), g, fan_in=HEADS * D_V),
                      router=(torch.randn(E, HID, device=dev, generator=gs) * HID ** -0.5).to(torch.bfloat16),
                      bias=torch.zeros(E, device=dev),
                      w13=fp8w((el, 2 * INTER, HID), g), w2=fp8w((el, HID, INTER), g),
                      sh13=fp8w((2 * INTER, HID), gs), sh2=fp8w((HID, INTER), gs)))
    a2a = AllToAll(TOPK, E, HID, t_loc * world, None, group=group)
    rows = world * a2a.cap * TOPK                                          # static bound of this rank's routed rows
    mp = (rows + el * 31) // 32 * 32
    hid_loc = torch.randn(t_loc, HID, device=dev, generator=g).to(torch.bfloat16)   # the layer input: this rank's token slice
    bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
    B = dict(res=torch.randn(t_loc, HID, device=dev, generator=g).to(torch.bfloat16), res2=bf(t_loc, HID), norm=bf(t_loc, HID),
             qkv_loc=bf(t_loc, QKV_A), qkv=bf(bs, QKV_A), qa=bf(bs, Q_LORA),
             qa8=torch.empty(bs, Q_LORA, dtype=torch.float8_e4m3fn, device=dev),
             qa8s=torch.empty((Q_LORA // 128, (bs + 3) // 4 * 4), dtype=torch.float32, device=dev).permute(-1, -2)[:bs],
             q=bf(bs, h, D_NOPE + D_ROPE), Qabs=bf(bs, h, KV_LORA + D_ROPE), attn_v=bf(bs, h * D_V), o=bf(bs, HID), norm2=bf(t_loc, HID),
             ex=torch.empty(el + 1, dtype=torch.int32, device=dev), xrows=torch.zeros(rows, HID, dtype=torch.bfloat16, device=dev),
             xq=torch.empty(rows, HID, dtype=torch.float8_e4m3fn, device=dev),
             xs=torch.empty((HID // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2),
             gate_up=bf(rows, 2 * INTER), dq=torch.empty(rows, INTER, dtype=torch.float8_e4m3fn, device=dev),
             ds=torch.empty((INTER // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2),
             down=bf(rows, HID), moe=bf(t_loc, HID), sh_gu=bf(t_loc, 2 * INTER),
             sh_a8=torch.empty(t_loc, INTER, dtype=torch.float8_e4m3fn, device=dev),
             sh_as=torch.empty((INTER // 128, (t_loc + 3) // 4 * 4), dtype=torch.float32, device=dev).permute(-1, -2)[:t_loc],
             sh_out=bf(t_loc, HID), out=bf(t_loc, HID))
    alt = torch.cuda.Stream(device=dev)
    fuse_q_quant = os.environ.get("FLUENT_CFG4_C7_QUANT", "1") != "0"

    def attention(l, w):
        k_lora, k_scale, k_rope = wl["caches"][l]
        pages = wl["pages"]
        q = B["q"]
        if USE_A2:     # bmm + RoPE + K5 + K4: one launch, the bf16 absorbed query is never written
            qn, qs, qr = fm.absorb_rope_quant(q, w["w_kc"], positions, cache, latent_cache=B["qkv"][:, Q_LORA:], k_lora_cache=k_lora,
                                              k_lora_scale_cache=k_scale, k_rope_cache=k_rope, indices=wl["out_loc"], is_neox=False)
        else:          # the model code's own sequence (deepseek_v2.py:830-861) + FlashMLABackend.forward_decode's two quantisers
            Q = B["Qabs"]
            bmm(q[..., :D_NOPE].transpose(0, 1), w["w_kc"], out=Q[..., :KV_LORA].transpose(0, 1))
            K = B["qkv"][:, Q_LORA:].unsqueeze(1)
            apply_rope_with_cos_sin_cache_inplace(positions, q[..., D_NOPE:], K[..., KV_LORA:], D_ROPE, cache, is_neox=False,
                                                  output_q_rope=Q[..., KV_LORA:])
            fm.quantize_and_cache_k(K.contiguous(), k_lora, k_scale, k_rope, wl["out_loc"], KV_LORA)
            qn, qs, qr = fm.quantize_ckv_per_token_head(Q.view(bs, 1, h, KV_LORA + D_ROPE), KV_LORA)
        B["qn"], B["qs"], B["qr"] = qn, qs, qr
        o, _ = fm.flash_mla_ckv_fp8_per_token(qn.view(bs, 1, h, KV_LORA), qr.view(bs, 1, h, D_ROPE), k_lora.view(pages, 64, 1, KV_LORA),
                                              k_rope.view(pages, 64, 1, D_ROPE), qs.view(bs, 1, h, 1), k_scale.view(pages, 64, 1, 1),
                                              wl["block_table"], wl["seqlens"], KV_LORA, meta, ns, bench.SCALE, True)
        B["mla_o"] = o
        bmm(o.view(bs, h, KV_LORA).transpose(0, 1), w["w_vc"], out=B["attn_v"].view(bs, h, D_V).transpose(0, 1))

    # ---- the layer as named stages (bench.py times the whole step as one hipGraph and, eagerly, every stage on its own) ----
    def st_input_norm(l, w):
        # residual add + input norm on the local slice (no exchange); the q_a / kv_a projection runs on the slice, its rows travel
        comm.trtllm_allreduce_fusion(allreduce_in=hid_loc, world_size=1, world_rank=0, token_num=t_loc, hidden_dim=HID,
                                     workspace_ptrs=ws_one, pattern_code=comm.AllReduceFusionPattern.kARResidualRMSNorm,
                                     residual_in=B["res"], residual_out=B["res2"], norm_out=B["norm"], rms_gamma=w["gamma1"], rms_eps=1e-6)

    def st_qkv_a(l, w):
        q8, s8 = per_token_group_quant_fp8(B["norm"], column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt((q8, s8), w["qkv_a"], B["qkv_loc"])

    def st_c7(l, w):
        # C7: all-gather + dual RMSNorm (+ 1x128 quant of the normed q_a): forward_with_allgather_fusion, deepseek_v2.py:803-810
        comm.trtllm_allgather_fusion(allgather_in=B["qkv_loc"], world_size=world, world_rank=rank, hidden_dim=QKV_A, workspace_ptrs=ws_tp,
                                     num_token_current_rank=t_loc, allgather_out=B["qkv"], num_token_all_group=bs,
                                     pattern_code=(comm.AllGatherFusionPattern.kAllGatherfusedRMSFP8BlockWiseQuant if fuse_q_quant
                                                   else comm.AllGatherFusionPattern.kAllGatherfusedRMS),
                                     x_norm_out=B["qa"], quant_out=B["qa8"] if fuse_q_quant else None,
                                     scale_out=B["qa8s"] if fuse_q_quant else None, x_rms_gamma=w["gamma_q"], y_rms_gamma=w["gamma_kv"],
                                     x_rms_eps=1e-6, y_rms_eps=1e-6, q_lora_rank=Q_LORA, kv_lora_rank=KV_LORA, qk_rope_head_dim=D_ROPE)

    def st_q_b(l, w):
        qa = (B["qa8"], B["qa8s"]) if fuse_q_quant else per_token_group_quant_fp8(B["qa"], column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt(qa, w["q_b"], B["q"].view(bs, h * (D_NOPE + D_ROPE)))

    def st_o_proj(l, w):
        oq, os_ = per_token_group_quant_fp8(B["attn_v"], column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt((oq, os_), w["o"], B["o"])

    def st_c6(l, w):
        # C6: reduce-scatter of the o_proj partial sums + residual + post-attention norm on the rank's slice
        comm.trtllm_reducescatter_fusion(reducescatter_in=B["o"], world_size=world, world_rank=rank, token_num=bs, hidden_dim=HID,
                                         workspace_ptrs=ws_tp, num_token_current_rank=t_loc,
                                         pattern_code=comm.ReduceScatterFusionPattern.kRSResidualRMSNorm, residual_in=B["res2"],
                                         residual_out=B["res"], norm_out=B["norm2"], rms_gamma=w["gamma2"], rms_eps=1e-6)

    def st_shared(l, w):
        hq, hs = per_token_group_quant_fp8(B["norm2"], column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt((hq, hs), w["sh13"], B["sh_gu"])
        silu_and_mul_fuse_block_quant(B["sh_gu"], B["sh_as"], B["sh_a8"])
        deep_gemm.gemm_fp8_fp8_bf16_nt((B["sh_a8"], B["sh_as"]), w["sh2"], B["sh_out"])

    def st_router(l, w):
        logits = flashinfer.dsv3_router_gemm(B["norm2"], w["router"], out_dtype=torch.float32)
        tw, ti = flashinfer.moe_fused_gate(logits, w["bias"], N_GROUP, TOPK_GROUP, TOPK, routed_scaling_factor=2.5)
        B["logits"], B["topk_w"], B["topk_ids"] = logits, tw, ti

    def st_dispatch(l, w):
        # (the routing weights travel in the dispatch message: one all-to-all per direction)
        a2a.dispatch(out_exclusive_sum=B["ex"], out_expert_x=B["xrows"], dp_x=B["norm2"], indices=B["topk_ids"], num_global_tokens=bs,
                     weights=B["topk_w"])

    def st_experts(l, w):
        flashinfer.quantization.quant_1x128(B["xrows"], B["xq"], B["xs"], B["ex"], el, (rows + 3) // 4 * 4, mp, HID)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((B["xq"], B["xs"]), w["w13"], B["gate_up"], B["ex"], use_pdl=True)
        act = silu(B["gate_up"], B["ex"], rows)
        flashinfer.quantization.quant_1x128(act, B["dq"], B["ds"], B["ex"], el, (rows + 3) // 4 * 4, mp, INTER)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((B["dq"], B["ds"]), w["w2"], B["down"], B["ex"], use_pdl=True)

    def st_combine(l, w):
        a2a.combine(out_tokens=B["moe"], weights=B["topk_w"], expert_y=B["down"], num_global_tokens=bs)

    def st_add(l, w):
        torch.add(B["moe"], B["sh_out"], out=B["out"])        # final_hidden_states = routed + shared (deepseek_v2.py:342-345)

    pre = [("input_norm (C5)", st_input_norm), ("qkv_a_proj (Q2 + G4, N=2112)", st_qkv_a), ("gather + dual norm + quant (C7)", st_c7),
           ("q_b_proj (G4)", st_q_b), ("query side + K1 + bmm_v (A2/K5/K4, K1, B2)", attention), ("o_proj (Q2 + G4)", st_o_proj),
           ("reduce-scatter + norm (C6)", st_c6)]
    routed = [("router (B3 + R1)", st_router), ("ep_dispatch (C1)", st_dispatch), ("routed experts (Q1, G1 w13, SiLU*mul, Q1, G1 w2)", st_experts),
              ("ep_combine (C2)", st_combine)]

    def layer(l):
        w = W[l]
        for _, f in pre:
            f(l, w)
        # ---- MoE: the shared expert on the second stream beside router + dispatch + routed experts + combine ----
        cur = torch.cuda.current_stream()
        alt.wait_stream(cur)
        with torch.cuda.stream(alt):
            st_shared(l, w)
        for _, f in routed:
            f(l, w)
        cur.wait_stream(alt)
        st_add(l, w)
        return B["out"]

    def step():
        for l in range(layers):
            layer(l)

    def stage_times(reps=5):
        """ms per stage of layer 0, each stage alone: `reps` back-to-back calls captured in ONE hipGraph and replayed between HIP events
        (eager launches of a three-kernel stage are CPU-bound, ~120 us per call; a stage that refuses capture is timed eagerly and
        marked).  The step itself is timed as one hipGraph by the caller — the sum of the stages is NOT the layer time: no overlap
        between stages here, the shared expert runs beside the routed experts in the step."""
        out = {}
        w = W[0]
        for name, f in pre + routed + [("shared expert (Q2, G4, A1, G4) [second stream in the step]", st_shared), ("routed + shared", st_add)]:
            f(0, w)
            torch.cuda.synchronize()
            gr = None
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    f(0, w)
                torch.cuda.current_stream().wait_stream(side)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(reps):
                        f(0, w)
                gr.replay()
                torch.cuda.synchronize()
            except Exception:
                gr = None
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if gr is not None:
                gr.replay()
            else:
                for _ in range(reps):
                    f(0, w)
            e1.record()
            torch.cuda.synchronize()
            out[name if gr is not None else name + " [eager]"] = round(e0.elapsed_time(e1) / reps, 4)
        return out

    kv_bytes = bench.algorithmic_bytes(bs, seq, h, 1)
    w_bytes = sum(W[0][k][0].numel() for k in ("qkv_a", "q_b", "o", "w13", "w2", "sh13", "sh2")) + 2 * (W[0]["w_kc"].numel() + W[0]["w_vc"].numel())
    xgmi = {  # bytes this rank SENDS per layer (bf16 rows), by collective
        "allgather": (world - 1) * t_loc * QKV_A * 2, "reducescatter": (world - 1) * t_loc * HID * 2,
        "ep_dispatch": (world - 1) * a2a.cap * (HID * 2 + TOPK * 8),   # row + top_k ids + top_k weights in the row tail
        "ep_combine": (world - 1) * a2a.cap * HID * 2}
    info = dict(bs=bs, seq=seq, heads_per_rank=h, experts_per_rank=el, tokens_per_rank=t_loc, routed_row_bound=rows,
                ep_slab_rows_per_peer=a2a.cap, kv_bytes_per_layer=kv_bytes, weight_bytes_per_layer=w_bytes,
                xgmi_send_bytes_per_layer=xgmi, query_side="A2 one launch (bmm + RoPE + K5 + K4)" if USE_A2 else "bmm, RoPE, K5, K4 (four launches)",
                data_connected=True, shared_expert="second stream",
                comm_route={"allgather / reducescatter (<= 1024 tokens)": "one-shot peer-mapped kernel" if getattr(h_tp[0], "oneshot", None) is not None
                            else "RCCL collective + fused kernel", "ep_dispatch / ep_combine": a2a.comm_route + ", ONE message each (ids + weights in the slab-row tail)"})
    # everything a checker needs to recompute the layer from its inputs (tests/test_cfg4_gpu.py); not used by the bench
    info["_state"] = dict(W=W, B=B, wl=wl, hid_loc=hid_loc, meta=meta, ns=ns, res_in=B["res"].clone(), positions=positions, cache=cache)
    info["_layer"] = layer
    info["_stage_times"] = stage_times
    return step, info
