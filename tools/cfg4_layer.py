"""BASELINE config 4 as a multi-rank step (bench.py --mode cfg4): attention-TP + EP MoE of a DeepSeek-V3 decoder layer with the
path's real collectives, one process per GPU.

Per layer, on every rank of a `world`-rank job (bs=256 requests, seq=8192, hidden 7168, 256 experts top-8, 128 heads):
  residual add + input RMSNorm on the rank's token slice (C5 kernel, no exchange: the hidden state stays reduce-scattered
  between layers, decoder_comm_manager.py:152-219 RSAG)
  -> all-gather of the normed rows (C7 kAllGather; RCCL)                                   [bs/world -> bs rows]
  -> 1x128 quant + q_a/kv_a projection (dense fp8 GEMM) -> K5 store + K4 quantise-q + K1 MLA decode over the rank's
     128/world heads, all bs requests (latent KV replicated under attention-TP)
  -> quant + o_proj (dense fp8 GEMM, partial sums over the TP group)
  -> C6: reduce-scatter + residual + post-attention RMSNorm (RCCL exchange + fused kernel) [bs -> bs/world rows]
  -> router logits (B1 bf16 MFMA GEMM) + R1 moe_fused_gate -> EP dispatch (eps.fast_ep.AllToAll: RCCL all-to-all of
     token-once-per-peer slabs) -> quant_1x128 -> grouped w13 -> SiLU*mul -> quant_1x128 -> grouped w2 -> EP combine (RCCL)
Synthetic random weights / activations (values do not matter for timing; the attention input q is synthetic as in the MLA
bench: the absorbed-q projection is outside this repo).  Everything has static shapes: one hipGraph per step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))

BS, SEQ, HEADS, HID, INTER, E, TOPK = 256, 8192, 128, 7168, 2048, 256, 8
QKV_A = 2176            # q_lora 1536 + kv_lora 512 + rope 64 = 2112, padded to the 128-wide weight blocks
N_GROUP, TOPK_GROUP = 8, 4


def build(dev, world, rank, group, layers, seq=SEQ, bs=BS, seed=0):
    """-> (step, info): `step()` runs `layers` decoder layers of this rank's share; `info` names the static sizes."""
    import bench
    import deep_gemm
    import flash_mla_fp8 as fm
    import flashinfer
    import flashinfer.comm as comm
    from eps.executor import silu
    from eps.fast_ep import AllToAll
    from fluent_mi355.gemm import per_token_group_quant_fp8

    if HEADS % world or E % world or bs % world:
        raise SystemExit(f"cfg4: world {world} must divide heads {HEADS}, experts {E} and bs {bs}")
    h, el, t_loc = HEADS // world, E // world, bs // world
    g = torch.Generator(device=dev).manual_seed(seed + 17 * rank)

    def fp8w(*shape):
        b = torch.randint(0, 255, shape, device=dev, generator=g, dtype=torch.int16)
        return torch.where((b & 0x7F) == 0x7F, b - 1, b).to(torch.uint8).view(torch.float8_e4m3fn)

    def ws(*shape):
        return torch.rand(*shape, device=dev, generator=g) * 1e-2

    wl = bench.build_workload(dev, layers, bs, seq, h, seed=seed + 3)      # same requests on every rank (TP): same seed
    meta, ns = fm.get_mla_metadata(wl["seqlens"], h, 1)
    h_tp, ws_tp = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(rank, world, bs, HID, group=group)
    _, ws_one = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(0, 1, bs, HID)        # the exchange-free norm
    W = []
    for _ in range(layers):
        W.append(dict(gamma1=torch.ones(HID, dtype=torch.bfloat16, device=dev), gamma2=torch.ones(HID, dtype=torch.bfloat16, device=dev),
                      qkv_a=(fp8w(QKV_A, HID), ws(QKV_A // 128, HID // 128)),
                      o=(fp8w(HID, h * 128), ws(HID // 128, h * 128 // 128)),
                      router=(torch.randn(E, HID, device=dev, generator=g) * 0.02).to(torch.bfloat16),
                      bias=torch.zeros(E, device=dev),
                      w13=(fp8w(el, 2 * INTER, HID), ws(el, 2 * INTER // 128, HID // 128)),
                      w2=(fp8w(el, HID, INTER), ws(el, HID // 128, INTER // 128))))
    a2a = AllToAll(TOPK, E, HID, t_loc * world, None, group=group)
    rows = world * a2a.cap * TOPK                                          # static bound of this rank's routed rows
    mp = (rows + el * 31) // 32 * 32
    hid_loc = torch.randn(t_loc, HID, device=dev, generator=g).to(torch.bfloat16)   # the layer input: this rank's token slice
    attn_o = torch.randn(bs, h * 128, device=dev, generator=g).to(torch.bfloat16)   # stands in for the absorbed-V output
    B = dict(res=torch.randn(t_loc, HID, device=dev, generator=g).to(torch.bfloat16), res2=torch.empty(t_loc, HID, dtype=torch.bfloat16, device=dev),
             norm=torch.empty(t_loc, HID, dtype=torch.bfloat16, device=dev), full=torch.empty(bs, HID, dtype=torch.bfloat16, device=dev),
             qkv=torch.empty(bs, QKV_A, dtype=torch.bfloat16, device=dev), o=torch.empty(bs, HID, dtype=torch.bfloat16, device=dev),
             norm2=torch.empty(t_loc, HID, dtype=torch.bfloat16, device=dev),
             ex=torch.empty(el + 1, dtype=torch.int32, device=dev), xrows=torch.zeros(rows, HID, dtype=torch.bfloat16, device=dev),
             xq=torch.empty(rows, HID, dtype=torch.float8_e4m3fn, device=dev),
             xs=torch.empty((HID // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2),
             gate_up=torch.empty(rows, 2 * INTER, dtype=torch.bfloat16, device=dev),
             dq=torch.empty(rows, INTER, dtype=torch.float8_e4m3fn, device=dev),
             ds=torch.empty((INTER // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2),
             down=torch.empty(rows, HID, dtype=torch.bfloat16, device=dev), moe=torch.empty(t_loc, HID, dtype=torch.bfloat16, device=dev))

    def layer(l):
        w = W[l]
        # residual add + input norm on the local slice (no exchange), then the rows travel to every TP rank
        comm.trtllm_allreduce_fusion(allreduce_in=hid_loc, world_size=1, world_rank=0, token_num=t_loc, hidden_dim=HID,
                                     workspace_ptrs=ws_one, pattern_code=comm.AllReduceFusionPattern.kARResidualRMSNorm,
                                     residual_in=B["res"], residual_out=B["res2"], norm_out=B["norm"], rms_gamma=w["gamma1"], rms_eps=1e-6)
        comm.trtllm_allgather_fusion(allgather_in=B["norm"], world_size=world, world_rank=rank, hidden_dim=HID, workspace_ptrs=ws_tp,
                                     num_token_current_rank=t_loc, allgather_out=B["full"], num_token_all_group=bs,
                                     pattern_code=comm.AllGatherFusionPattern.kAllGather)
        q8, s8 = per_token_group_quant_fp8(B["full"], column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt((q8, s8), w["qkv_a"], B["qkv"])
        B["mla_o"], _ = bench.layer_call(fm, wl, l, meta, ns)                          # K5 + K4 + K1
        oq, os_ = per_token_group_quant_fp8(attn_o, column_major_scales=True)
        deep_gemm.gemm_fp8_fp8_bf16_nt((oq, os_), w["o"], B["o"])
        # C6: reduce-scatter of the o_proj partial sums + residual + post-attention norm on the rank's slice
        comm.trtllm_reducescatter_fusion(reducescatter_in=B["o"], world_size=world, world_rank=rank, token_num=bs, hidden_dim=HID,
                                         workspace_ptrs=ws_tp, num_token_current_rank=t_loc,
                                         pattern_code=comm.ReduceScatterFusionPattern.kRSResidualRMSNorm, residual_in=B["res2"],
                                         residual_out=B["res"], norm_out=B["norm2"], rms_gamma=w["gamma2"], rms_eps=1e-6)
        logits = flashinfer.dsv3_router_gemm(B["norm2"], w["router"], out_dtype=torch.float32)   # B1: bf16 MFMA kernel (csrc/bmm_bf16.hip)
        tw, ti = flashinfer.moe_fused_gate(logits, w["bias"], N_GROUP, TOPK_GROUP, TOPK, routed_scaling_factor=2.5)
        B["logits"], B["topk_w"], B["topk_ids"] = logits, tw, ti
        # (the routing weights travel in the dispatch message: one all-to-all per direction)
        a2a.dispatch(out_exclusive_sum=B["ex"], out_expert_x=B["xrows"], dp_x=B["norm2"], indices=ti, num_global_tokens=bs, weights=tw)
        flashinfer.quantization.quant_1x128(B["xrows"], B["xq"], B["xs"], B["ex"], el, (rows + 3) // 4 * 4, mp, HID)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((B["xq"], B["xs"]), w["w13"], B["gate_up"], B["ex"], use_pdl=True)
        act = silu(B["gate_up"], B["ex"], rows)
        flashinfer.quantization.quant_1x128(act, B["dq"], B["ds"], B["ex"], el, (rows + 3) // 4 * 4, mp, INTER)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((B["dq"], B["ds"]), w["w2"], B["down"], B["ex"], use_pdl=True)
        a2a.combine(out_tokens=B["moe"], weights=tw, expert_y=B["down"], num_global_tokens=bs)
        return B["moe"]

    def step():
        for l in range(layers):
            layer(l)

    kv_bytes = bench.algorithmic_bytes(bs, seq, h, 1)
    w_bytes = sum(W[0][k][0].numel() for k in ("qkv_a", "o", "w13", "w2"))
    xgmi = {  # bytes this rank SENDS per layer (bf16 rows), by collective
        "allgather": (world - 1) * t_loc * HID * 2, "reducescatter": (world - 1) * t_loc * HID * 2,
        "ep_dispatch": (world - 1) * a2a.cap * (HID * 2 + TOPK * 8),   # row + top_k ids + top_k weights in the row tail
        "ep_combine": (world - 1) * a2a.cap * HID * 2}
    info = dict(bs=bs, seq=seq, heads_per_rank=h, experts_per_rank=el, tokens_per_rank=t_loc, routed_row_bound=rows,
                ep_slab_rows_per_peer=a2a.cap, kv_bytes_per_layer=kv_bytes, weight_bytes_per_layer=w_bytes,
                xgmi_send_bytes_per_layer=xgmi,
                comm_route={"allgather / reducescatter (<= 1024 tokens)": "one-shot peer-mapped kernel" if getattr(h_tp[0], "oneshot", None) is not None
                            else "RCCL collective + fused kernel", "ep_dispatch / ep_combine": "ONE RCCL all_to_all_single each (ids + weights in the slab-row tail)"})
    # everything a checker needs to recompute the layer from its inputs (tests/test_cfg4_gpu.py); not used by the bench
    info["_state"] = dict(W=W, B=B, wl=wl, hid_loc=hid_loc, attn_o=attn_o, meta=meta, ns=ns, res_in=B["res"].clone())
    return step, info
