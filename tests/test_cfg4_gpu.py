"""BASELINE config 4 end to end on one GPU (world 1): ONE DeepSeek-V3 decoder layer exactly as `bench.py --mode cfg4` runs
it (tools/cfg4_layer.py: input norm -> all-gather -> q_a/kv_a fp8 GEMM -> K5 + K4 + K1 -> o_proj fp8 GEMM -> C6 norm ->
router -> EP dispatch -> quant -> grouped w13 -> SiLU*mul -> quant -> grouped w2 -> EP combine), every stage compared with
the ORACLE composition norm_ref -> gemm_ref -> mla_ref -> router_ref -> MoE (gemm_ref).  Full model dimensions (hidden
7168, 256 experts x [4096, 7168] / [7168, 2048], 128 heads); batch and context are small so that the CPU oracle finishes in
seconds, and the MoE is checked on sampled tokens (each touches 8 experts: 48 expert weight pairs dequantised on the CPU)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _rel_mae(x, ref):
    x, ref = x.double().cpu(), ref.double().cpu()
    return float((x - ref).abs().mean() / ref.abs().mean().clamp_min(1e-30))


def test_cfg4_layer_world1_matches_the_oracle_composition():
    import cfg4_layer as L
    from oracle import gemm_ref, mla_ref, norm_ref, router_ref

    dev = torch.device("cuda:0")
    bs, seq = 32, 300
    step, info = L.build(dev, world=1, rank=0, group=None, layers=1, seq=seq, bs=bs, seed=5)
    st = info["_state"]
    W, B, wl = st["W"][0], st["B"], st["wl"]
    res_in = st["res_in"]                       # B["res"] is overwritten by the C6 stage: the input copy
    k_before = [t.clone() for t in wl["caches"][0]]
    step()
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu()

    # ---- 1. residual add + input RMSNorm (C5 kernel at world 1) ----
    n1, r1 = norm_ref.fused_add_rmsnorm(cpu(st["hid_loc"]).unsqueeze(0), None, cpu(res_in), cpu(W["gamma1"]), 1e-6)
    assert torch.equal(cpu(B["res2"]).view(torch.int16), r1.view(torch.int16))
    assert float((cpu(B["norm"]).float() - n1.float()).abs().max()) <= 2 ** -7 * float(n1.float().abs().max())   # <= 1 bf16 ulp
    assert torch.equal(B["full"], B["norm"])    # world 1: the all-gather is a copy

    # ---- 2. q_a / kv_a projection: 1x128 quant + dense fp8 GEMM ----
    a_q, a_s = gemm_ref.per_token_group_quant_fp8(cpu(B["full"]))
    qkv = gemm_ref.block_fp8_matmul(a_q, cpu(W["qkv_a"][0]), a_s, cpu(W["qkv_a"][1]))
    assert _rel_mae(B["qkv"], qkv) < 1e-3       # the reference's own block-fp8 threshold (test_block_fp8.py:185-189)

    # ---- 3. K5 + K4 + K1 over all 128 heads ----
    k_lora, k_scale, k_rope = [cpu(t) for t in k_before]
    mla_ref.quantize_and_cache_k(cpu(wl["k_new"]), k_lora, k_scale, k_rope, cpu(wl["out_loc"]))
    for got, want in zip(wl["caches"][0], (k_lora, k_scale, k_rope)):
        assert torch.equal(cpu(got).view(torch.uint8), want.view(torch.uint8))           # K5 bytes
    qn, qs, qr = mla_ref.quantize_ckv_per_token_head(cpu(wl["q"]), 512)
    pages = wl["pages"]
    ref_o, _ = mla_ref.mla_decode_fp8_per_token(qn, qs, qr, k_lora.view(pages, 64, 1, 512), k_scale.view(pages, 64, 1, 1),
                                                k_rope.view(pages, 64, 1, 64), cpu(wl["block_table"]), cpu(wl["seqlens"]),
                                                192 ** -0.5, True)
    from test_mla_gpu import K1_REL_MAE_BOUND
    assert _rel_mae(B["mla_o"], ref_o) < K1_REL_MAE_BOUND

    # ---- 4. o_proj ----
    o_q, o_s = gemm_ref.per_token_group_quant_fp8(cpu(st["attn_o"]))
    o = gemm_ref.block_fp8_matmul(o_q, cpu(W["o"][0]), o_s, cpu(W["o"][1]))
    assert _rel_mae(B["o"], o) < 1e-3

    # ---- 5. C6 at world 1: sum of one piece + residual + post-attention norm (from the KERNEL's o: stage errors do not stack) ----
    n2, r2 = norm_ref.fused_add_rmsnorm(cpu(B["o"]).unsqueeze(0), None, r1, cpu(W["gamma2"]), 1e-6)
    assert torch.equal(cpu(B["res"]).view(torch.int16), r2.view(torch.int16))
    assert float((cpu(B["norm2"]).float() - n2.float()).abs().max()) <= 2 ** -7 * float(n2.float().abs().max())

    # ---- 6. router: selection on the kernel's own logits (the router GEMM is a library GEMM), ids as sets ----
    rw, rid = router_ref.biased_grouped_topk(cpu(B["logits"]).numpy(), cpu(W["bias"]).numpy(), L.N_GROUP, L.TOPK_GROUP, L.TOPK,
                                             routed_scaling_factor=2.5)
    gid, gw = cpu(B["topk_ids"]).numpy(), cpu(B["topk_w"]).numpy()
    same = 0
    for t in range(bs):
        if set(gid[t].tolist()) != set(rid[t].tolist()):
            continue   # (float32 sigmoid on the device vs float64-rounded on the host: 1 ulp can swap two near-equal picks)
        same += 1
        order_g, order_r = np.argsort(gid[t]), np.argsort(rid[t])
        assert np.allclose(gw[t][order_g], rw[t][order_r], rtol=0, atol=2e-6)
    assert same >= bs - 1

    # ---- 7. MoE (dispatch -> quant -> w13 -> SiLU*mul -> quant -> w2 -> combine) on sampled tokens ----
    x = cpu(B["norm2"])
    for t in (0, 7, 13, 21, 30, 31):
        ids = torch.from_numpy(gid[t].astype(np.int64))
        ref_row = gemm_ref.moe_fp8_block(
            x[t:t + 1], cpu(W["w13"][0][ids]), cpu(W["w2"][0][ids]), cpu(W["w13"][1][ids]), cpu(W["w2"][1][ids]),
            torch.from_numpy(gw[t:t + 1].copy()), torch.arange(L.TOPK).view(1, -1))
        assert _rel_mae(B["moe"][t:t + 1], ref_row) < 2e-2, t       # the reference's MoE threshold (test_block_fp8.py:310-314)
