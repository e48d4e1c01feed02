"""CPU: pins oracle/*.py against golden vectors produced by the REAL reference (oracle/gen_golden.py)."""
import numpy as np
import torch

from helpers import bf16_from_u16, fp8_from_u8, load_golden, rel_mae
from oracle import gemm_ref, mla_ref


def test_kv_quant_bit_exact_vs_reference():
    g = load_golden("kv_quant_per_token.npz")
    key = bf16_from_u16(g["key"])
    loc = torch.from_numpy(g["loc"])
    k_lora = torch.zeros(g["k_lora"].shape, dtype=torch.uint8)
    k_scale = torch.zeros(g["k_scale"].shape, dtype=torch.float32)
    k_rope = torch.zeros(g["k_rope"].shape, dtype=torch.bfloat16)
    mla_ref.quantize_and_cache_k(key, k_lora, k_scale, k_rope, loc)
    idx = loc.long()
    assert np.array_equal(k_lora.numpy()[idx], g["k_lora"][idx])
    assert np.array_equal(k_scale.numpy()[idx].view(np.uint32), g["k_scale"][idx].view(np.uint32))
    assert np.array_equal(k_rope.view(torch.int16).numpy().view(np.uint16)[idx], g["k_rope"][idx])


def test_dequant_bit_exact_vs_reference():
    g = load_golden("kv_quant_per_token.npz")
    lora, rope = mla_ref.dequantize_ckv_fused_indexed(torch.from_numpy(g["k_lora"]), bf16_from_u16(g["k_rope"]),
                                                      torch.from_numpy(g["k_scale"]), torch.from_numpy(g["gather"]))
    assert torch.equal(lora.view(torch.int16), bf16_from_u16(g["lora_deq"]).view(torch.int16))
    assert torch.equal(rope.view(torch.int16), bf16_from_u16(g["rope_deq"]).view(torch.int16))


def test_alloc_page_arithmetic_bit_exact():
    g = load_golden("kv_alloc.npz")
    free = list(g["free_slots"])
    r2p = torch.zeros(g["req_to_page"].shape, dtype=torch.int32)
    for i, (req, need, alloced) in enumerate(g["steps"]):
        page_num = (alloced + 63) // 64
        remain_after = max(0, need - (page_num * 64 - alloced))
        n_new = (remain_after + 63) // 64
        loc = mla_ref.alloc_kv_loc(r2p[req], free[:n_new], int(need), int(alloced))
        free = free[n_new:]
        assert np.array_equal(loc.numpy(), g[f"loc{i}"]), f"step {i}"
    assert np.array_equal(r2p.numpy(), g["req_to_page"])
    # token -> slot rule
    for req in range(r2p.shape[0]):
        assert mla_ref.kv_slot(r2p, req, 70) == int(r2p[req, 1]) * 64 + 6


def _native_case(name):
    g = load_golden(f"mla_torch_native_{name}.npz")
    H = int(g["H"])
    q = bf16_from_u16(g["q"]).view(-1, H, 576)
    kv = bf16_from_u16(g["kv_buffer_after"])
    return g, H, q, kv


def test_torch_native_decode_matches_reference():
    for name in ("cfg1", "ragged", "h128"):
        g, H, q, kv = _native_case(name)
        o = mla_ref.torch_native_decode(q, kv, torch.from_numpy(g["req_to_token"]), torch.arange(q.shape[0]),
                                        torch.from_numpy(g["seq_lens"]), float(g["scaling"]))
        ref = bf16_from_u16(g["o"])
        # same torch, same loop: normally bit-identical; allow bf16 rounding for other host CPUs' SDPA kernels
        assert (o.float() - ref.float()).abs().max() < 2e-2, name
        assert rel_mae(o, ref) < 2e-3, name


def test_exact_attention_oracle_matches_reference_backend():
    """mla_decode_exact (fp64, page-table gather) vs the reference's TorchNativeAttnBackend output."""
    for name in ("cfg1", "ragged", "h128"):
        g, H, q, kv = _native_case(name)
        bs = q.shape[0]
        o, lse = mla_ref.mla_decode_exact(q.view(bs, 1, H, 576).float(), kv.view(-1, 576).float(),
                                          torch.from_numpy(g["block_table"]), torch.from_numpy(g["seq_lens"]),
                                          float(g["scaling"]), 512, causal=True)
        ref = bf16_from_u16(g["o"]).view(bs, 1, H, 512)
        assert (o.float() - ref.float()).abs().max() < 2e-2, name  # reference computes in bf16
        assert torch.isfinite(lse).all()


def test_gemm_oracles_match_reference():
    g = load_golden("gemm_block_fp8.npz")
    xq, xs = gemm_ref.per_token_group_quant_fp8(bf16_from_u16(g["quant_x"]), 128)
    assert np.array_equal(xq.view(torch.uint8).numpy(), g["quant_q"])
    assert np.array_equal(xs.numpy().view(np.uint32), g["quant_s"].view(np.uint32))
    C = gemm_ref.block_fp8_matmul(fp8_from_u8(g["mm_A"]), fp8_from_u8(g["mm_B"]), torch.from_numpy(g["mm_As"]),
                                  torch.from_numpy(g["mm_Bs"]))
    assert rel_mae(C, bf16_from_u16(g["mm_C"])) < 1e-3
    act = gemm_ref.silu_and_mul(bf16_from_u16(g["silu_x"]))
    assert torch.equal(act.view(torch.int16), bf16_from_u16(g["silu_out"]).view(torch.int16))
    out = gemm_ref.moe_fp8_block(bf16_from_u16(g["moe_a"]), fp8_from_u8(g["moe_w1"]), fp8_from_u8(g["moe_w2"]),
                                 torch.from_numpy(g["moe_w1_s"]), torch.from_numpy(g["moe_w2_s"]),
                                 torch.from_numpy(g["moe_topk_w"]), torch.from_numpy(g["moe_topk_ids"]).long())
    assert rel_mae(out, bf16_from_u16(g["moe_out"])) < 2e-2


def test_grouped_gemm_variants_agree():
    g = torch.Generator().manual_seed(5)
    E, N, K = 3, 256, 256
    counts = [5, 0, 9]
    M = sum(counts)
    A = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    Aq, As = gemm_ref.per_token_group_quant_fp8(A, 128)
    W = ((torch.rand(E, N, K, generator=g) - 0.5) * 896).clamp(-448, 448).to(torch.float8_e4m3fn)
    Ws = torch.rand(E, N // 128, K // 128, generator=g) * 1e-2
    ex = torch.tensor([0, 5, 5, 14], dtype=torch.int32)
    o1 = gemm_ref.grouped_gemm_offset(Aq, As, W, Ws, ex)
    mi = torch.tensor([0] * 5 + [2] * 9)
    o2 = gemm_ref.grouped_gemm_contiguous(Aq, As, W, Ws, mi)
    assert torch.equal(o1, o2)
    Am = torch.zeros(E, 16, K, dtype=torch.float8_e4m3fn)
    Asm = torch.zeros(E, 16, K // 128)
    Am[0, :5], Asm[0, :5] = Aq[:5], As[:5]
    Am[2, :9], Asm[2, :9] = Aq[5:], As[5:]
    o3 = gemm_ref.grouped_gemm_masked(Am, Asm, W, Ws, torch.tensor(counts))
    assert torch.equal(o3[0, :5], o1[:5]) and torch.equal(o3[2, :9], o1[5:])


def test_rmsnorm_oracle_matches_reference_golden():
    """oracle.norm_ref.rmsnorm_native == RMSNorm.forward_native of the real reference (layernorm.py:88-112), bit for bit."""
    from oracle import norm_ref
    g = load_golden("rmsnorm_native.npz")
    for name in ("h7168", "q1536", "kv512"):
        w, x, r = (bf16_from_u16(g[f"{name}_{k}"]) for k in ("w", "x", "r"))
        assert torch.equal(norm_ref.rmsnorm_native(x, w, 1e-6), bf16_from_u16(g[f"{name}_y"]))
        y, ro = norm_ref.rmsnorm_native(x, w, 1e-6, r)
        assert torch.equal(y, bf16_from_u16(g[f"{name}_y_res"])) and torch.equal(ro, bf16_from_u16(g[f"{name}_r_out"]))
        # the fused statement with one piece and a residual is the same arithmetic
        y2, r2 = norm_ref.fused_add_rmsnorm(x.unsqueeze(0), None, r, w, 1e-6)
        assert torch.equal(y2, y) and torch.equal(r2, ro)


ROUTER_CASES = ("dsv3", "dsv3_noscale", "e64", "e128_g1")


def router_case(g, name):
    G, TG, K, on_out, npad = (int(v) for v in g[name + "_cfg"])
    return dict(logits=g[name + "_logits"], bias=g[name + "_bias"], G=G, TG=TG, K=K, on_out=bool(on_out),
                npad=None if npad < 0 else npad, scale=float(g[name + "_scale"][0]), w=g[name + "_w"], ids=g[name + "_ids"])


def assert_router_rows_equal(w, ids, w_ref, ids_ref, npad, tol=2e-6):
    """torch.topk(sorted=False) leaves the order inside a row unspecified: compare rows as {expert: weight} maps; padded
    rows (ids -1, topk.py:673-680) only by their ids."""
    o, o_ref = np.argsort(ids, 1, kind="stable"), np.argsort(ids_ref, 1, kind="stable")
    assert np.array_equal(np.take_along_axis(ids, o, 1), np.take_along_axis(ids_ref, o_ref, 1))
    live = slice(None) if npad is None else slice(0, npad)
    assert np.abs(np.take_along_axis(w, o, 1)[live] - np.take_along_axis(w_ref, o_ref, 1)[live]).max() < tol


def test_router_oracle_matches_reference_golden():
    """oracle.router_ref.biased_grouped_topk ≡ the reference's biased_grouped_topk_impl (moe/topk.py:596-663) run from its
    own source: same expert sets (bit-exact ids), weights within 2e-6 (float32 sigmoid)."""
    from oracle import router_ref

    g = load_golden("router_biased_grouped_topk.npz")
    for name in ROUTER_CASES:
        c = router_case(g, name)
        w, ids = router_ref.biased_grouped_topk(c["logits"], c["bias"], c["G"], c["TG"], c["K"], c["scale"], c["on_out"], c["npad"])
        assert_router_rows_equal(w, ids, c["w"], c["ids"], c["npad"])


TOPK_PLAIN_CASES = ("softmax_e128", "softmax_e8", "softmax_e256", "softmax_e96", "bias_longcat", "bias_e64")


def topk_plain_case(g, name):
    K, renorm = (int(v) for v in g[name + "_cfg"])
    return dict(logits=g[name + "_logits"], bias=g[name + "_bias"] if name + "_bias" in g.files else None, K=K, renorm=bool(renorm),
                w=g[name + "_w"], ids=g[name + "_ids"])


def test_plain_topk_router_oracle_matches_reference_golden():
    """oracle.router_ref.topk_plain ≡ the reference's fused_topk_native (moe/topk.py:73-91, behind flashinfer.topk_softmax) and
    fused_topk_bias (:51-70, behind flashinfer.routing_flash) run from their own source: same expert sets, weights within 2e-6."""
    from oracle import router_ref

    g = load_golden("router_topk_plain.npz")
    for name in TOPK_PLAIN_CASES:
        c = topk_plain_case(g, name)
        w, ids = router_ref.topk_plain(c["logits"], c["K"], c["renorm"], c["bias"])
        assert_router_rows_equal(w, ids, c["w"], c["ids"], None)


def test_rope_oracle_bit_exact_vs_reference_golden():
    """oracle.rope_ref.apply_rope ≡ DeepseekScalingRotaryEmbedding.forward_native (rotary_embedding.py:804-846) run from the
    reference's own source with its YaRN fp32 cache: bf16 bits identical, GPT-J and NeoX styles."""
    from oracle import rope_ref

    g = load_golden("rope_deepseek_yarn.npz")
    for name, neox in (("gptj", False), ("neox", True)):
        for t in ("q", "k"):
            out = rope_ref.apply_rope(g[name + "_pos"], g[f"{name}_{t}"], g[name + "_cache"], neox)
            assert np.array_equal(out, g[f"{name}_{t}_out"]), (name, t)


def kv_move_buffers(g, suffix=""):
    return [t for l in range(2) for t in (torch.from_numpy(g[f"lora{l}{suffix}"].copy()), torch.from_numpy(g[f"scale{l}{suffix}"].copy()),
                                          bf16_from_u16(g[f"rope{l}{suffix}"]).clone())]


def test_kv_move_oracle_bit_exact_vs_reference_golden():
    """oracle.mla_ref.move_kv_cache ≡ the reference's move_kv_cache_native (memory_pool.py:2039-2052) on overlapping
    source / target sets."""
    g = load_golden("kv_move.npz")
    bufs = kv_move_buffers(g)
    mla_ref.move_kv_cache(bufs, torch.from_numpy(g["tgt"]), torch.from_numpy(g["src"]))
    for got, want in zip(bufs, kv_move_buffers(g, "_out")):
        assert torch.equal(got.view(torch.uint8), want.view(torch.uint8))


def test_ep_scatter_gather_oracle_exact_vs_reference_triton_golden():
    """oracle.ep_ref ≡ the reference's Triton ep_scatter / ep_gather (deep_ep_executor.py:173-430) as run by Triton's CPU
    interpreter (program order): every output of the scatter bit-exact, the gather's fp32 sums bit-exact."""
    from oracle import ep_ref
    from oracle.rope_ref import bf16_to_f32

    g = load_golden("ep_scatter_gather.npz")
    start, out, outs, m_idx, oidx = ep_ref.ep_scatter(g["x"], g["xs"], g["topk"], g["padded"])
    assert np.array_equal(start, g["start_after"]) and np.array_equal(out, g["out"]) and np.array_equal(outs, g["outs"])
    assert np.array_equal(m_idx, g["m_idx"]) and np.array_equal(oidx, g["oidx"])
    got = ep_ref.ep_gather(bf16_to_f32(g["y"]), g["topk"], g["w"], g["oidx"])
    assert np.array_equal(got, g["gathered"])


def test_metadata_positions_statement_equals_the_two_walk_statement():
    """K3: the cost-axis formulation the one-walk kernel implements (mla_ref.get_mla_metadata_positions) is the SAME partition as
    mla_ref.get_mla_metadata on uniform, ragged, empty, zero-length and oversubscribed batches (fixed seed)."""
    import random
    rnd = random.Random(7)
    cases = [([4096] * 128, 128), ([4096] * 128, 256), ([1] * 160, 256), ([0, 5, 200, 0, 9000], 128), ([16384], 128), ([], 16), ([0], 8),
             ([0, 0, 0], 4), ([100000] * 3, 4), ([63, 64, 65, 4095, 4097] * 7, 32)]
    for _ in range(400):
        bs = rnd.choice([1, 2, 3, 7, 16, 50, 128, 300])
        mx = rnd.choice([1, 64, 130, 1000, 5000, 20000])
        lens = [rnd.choice([0, rnd.randint(0, mx), mx]) for _ in range(bs)]
        if rnd.random() < 0.3:
            lens = [mx] * bs
        cases.append((lens, rnd.choice([1, 2, 3, 8, 32, 64, 128, 256])))
    for lens, parts in cases:
        a, b = mla_ref.get_mla_metadata(lens, parts), mla_ref.get_mla_metadata_positions(lens, parts)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (lens[:8], len(lens), parts)


def test_metadata_statements_agree_property_based():
    """hypothesis: arbitrary small batches (zero-length requests included) and part counts — the cost-axis statement equals the two-walk statement, every
    request is covered exactly once, parts are contiguous and num_splits counts the parts that touch a request."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=300, deadline=None)
    @given(st.lists(st.integers(min_value=0, max_value=3000), min_size=0, max_size=40), st.sampled_from([1, 2, 3, 5, 8, 16, 32, 64, 128, 256]))
    def check(lens, parts):
        a, b = mla_ref.get_mla_metadata(lens, parts), mla_ref.get_mla_metadata_positions(lens, parts)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        meta, ns = a
        bs = len(lens)
        nt = [(L + 63) // 64 if L > 0 else 0 for L in lens]
        touched = [0] * bs
        prev_end = (0, 0)
        for p in range(parts):
            br, bt, er, et = (int(x) for x in meta[p, :4])
            if br >= bs:
                assert (er, et) == (bs, 0)
                continue
            assert (br, bt) == prev_end and (er, et) >= (br, bt)            # contiguous, monotone
            for r in range(br, min(er + (1 if et > 0 else 0), bs)):
                touched[r] += 1
            assert et == 0 or et < nt[er]
            prev_end = (er, et)
        assert prev_end == (bs, 0) or bs == 0
        assert [int(ns[r + 1] - ns[r]) for r in range(bs)] == touched
    check()
