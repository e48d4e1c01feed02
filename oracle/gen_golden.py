"""TEST INFRASTRUCTURE ONLY (build-container side).  Generates tests/golden/*.npz.

Runs the REAL reference (imported from /root/reference/python via oracle/_ref_import.py) on
seeded inputs and records inputs + outputs as data fixtures.  The fixtures pin oracle/*.py
(tests/test_oracle_golden.py) and are compared with the HIP path on the GPU box, where the
reference itself does not exist.  Re-run:  python -B oracle/gen_golden.py

Reference entry points exercised (file:line under /root/reference/python/sglang):
  srt/layers/attention/torch_native_backend.py:472-528,275-343   TorchNativeAttnBackend.forward_decode
  srt/mem_cache/memory_pool.py:854-882,817-839                   MLATokenToKVPool.set_kv_buffer / get_key_split_contiguous
  srt/mem_cache/allocator.py:60-102                              KVAllocator.alloc
  test/test_block_fp8.py:15-40,89-141,212-241                    native fp8 quant / block matmul / MoE oracles
  srt/layers/activation.py:58-60                                 SiluAndMul.forward_native
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

assert _ref_import.install(), "reference tree not present: golden vectors can only be generated in the build container"
import torch  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def bf(x):  # bf16 tensor -> uint16 numpy
    return x.contiguous().view(torch.int16).numpy().view(np.uint16)


def f8(x):  # fp8 tensor -> uint8 numpy
    return x.contiguous().view(torch.uint8).numpy()


def gen_mla_torch_native():
    from sglang.srt.layers.attention.torch_native_backend import TorchNativeAttnBackend
    from sglang.srt.layers.radix_attention import RadixAttention
    from sglang.srt.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool
    from sglang.srt.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    cases = {
        # BASELINE.json configs[0]: kv_lora=512, 16 heads, bs=1, seq=128
        "cfg1": dict(H=16, lens=[128], seed=0),
        # ragged: page-boundary lengths 1/63/64/65/130, scattered pages
        "ragged": dict(H=16, lens=[1, 63, 64, 65, 130], seed=1),
        "h128": dict(H=128, lens=[200, 77], seed=2),
    }
    for name, c in cases.items():
        H, lens = c["H"], c["lens"]
        g = torch.Generator().manual_seed(c["seed"])
        bs = len(lens)
        max_ctx = 256
        npages = sum((L + 63) // 64 for L in lens)
        size = (npages + 2) * 64
        pool = MLATokenToKVPool(size, model_dtype=torch.bfloat16, dtype=torch.bfloat16, quant_method="none",
                                kv_lora_rank=512, qk_rope_head_dim=64, layer_num=1, device="cpu",
                                enable_memory_saver=False, max_batch_size=bs, max_context_len=max_ctx,
                                page_size=64, rank=0, enable_alt_stream=False)
        r2t = ReqToTokenPool(bs, max_ctx, "cpu", False)
        layer = RadixAttention(H, 576, 192 ** -0.5, num_kv_heads=1, layer_id=0, v_head_dim=512)
        perm = (torch.randperm(npages, generator=g) + 1).tolist()  # page 0 = padding page
        block_table = torch.zeros(bs, max_ctx // 64, dtype=torch.int32)
        kv_all = torch.randn(size + 64, 1, 576, generator=g).to(torch.bfloat16)
        pool.kv_buffer[0].copy_(kv_all)
        pi = 0
        out_loc = []
        for b, L in enumerate(lens):
            for t in range(L):
                if t % 64 == 0:
                    block_table[b, t // 64] = perm[pi]
                    pi += 1
                r2t.req_to_token[b, t] = int(block_table[b, t // 64]) * 64 + t % 64
            out_loc.append(int(r2t.req_to_token[b, L - 1]))
        q = torch.randn(bs, H * 576, generator=g).to(torch.bfloat16)
        k_new = torch.randn(bs, 1, 576, generator=g).to(torch.bfloat16)
        fb = ForwardBatch(forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=torch.zeros(bs, dtype=torch.int64),
                          req_pool_indices=torch.arange(bs), seq_lens=torch.tensor(lens, dtype=torch.int64),
                          seq_lens_sum=sum(lens), out_cache_loc=torch.tensor(out_loc, dtype=torch.int64),
                          req_to_token_pool=r2t, token_to_kv_pool=pool)
        be = TorchNativeAttnBackend(SimpleNamespace(device="cpu"))
        o = be.forward_decode(q, k_new, k_new[..., :512], layer, fb, save_kv_cache=True)
        np.savez_compressed(os.path.join(OUT, f"mla_torch_native_{name}.npz"),
                            H=H, seq_lens=np.array(lens, np.int32), scaling=np.float64(192 ** -0.5),
                            q=bf(q), k_new=bf(k_new), out_cache_loc=np.array(out_loc, np.int32),
                            kv_buffer_after=bf(pool.kv_buffer[0]), block_table=block_table.numpy(),
                            req_to_token=r2t.req_to_token.numpy().astype(np.int32), o=bf(o))
        print("mla_torch_native", name, tuple(o.shape))


def gen_kv_quant():
    from sglang.srt.layers.radix_attention import RadixAttention
    from sglang.srt.mem_cache.memory_pool import MLATokenToKVPool

    g = torch.Generator().manual_seed(10)
    n = 96
    size = 4 * 64
    pool = MLATokenToKVPool(size, model_dtype=torch.bfloat16, dtype=torch.float8_e4m3fn,
                            quant_method="per_token_head", kv_lora_rank=512, qk_rope_head_dim=64, layer_num=1,
                            device="cpu", enable_memory_saver=False, max_batch_size=4, max_context_len=256,
                            page_size=64, rank=0, enable_alt_stream=False)
    for t in pool.kv_buffer[0]:
        t.view(torch.uint8).zero_() if t.dtype != torch.float32 and t.dtype != torch.bfloat16 else t.zero_()
    key = torch.randn(n, 1, 576, generator=g)
    key[0] = 0.0                        # all-zero row -> clamp(1e-26) path
    key[1] *= 1e-30                     # subnormal-ish amax
    key[2] *= 1e4                       # large values
    key[3, 0, :512] = 0.0               # zero latent, non-zero rope
    key[4] *= 3e-3
    key[5, 0, 7] = 448.0                # exact fp8 max
    key[6, 0, :512] = torch.linspace(-1, 1, 512)  # ties
    key = key.to(torch.bfloat16)
    loc = (torch.randperm(size, generator=g)[:n] + 64).to(torch.int64)  # skip padding page 0
    layer = RadixAttention(16, 576, 192 ** -0.5, num_kv_heads=1, layer_id=0, v_head_dim=512)
    pool.set_kv_buffer(layer, loc, key, key[..., :512])
    k_lora, k_scale, k_rope = pool.kv_buffer[0]
    gather = torch.cat([loc[:40], loc[:8]])
    lora_deq, rope_deq = pool.get_key_split_contiguous(0, gather)
    np.savez_compressed(os.path.join(OUT, "kv_quant_per_token.npz"), key=bf(key), loc=loc.numpy().astype(np.int32),
                        k_lora=f8(k_lora), k_scale=k_scale.numpy(), k_rope=bf(k_rope),
                        gather=gather.numpy().astype(np.int32), lora_deq=bf(lora_deq), rope_deq=bf(rope_deq))
    print("kv_quant", tuple(k_lora.shape))


def gen_alloc():
    from sglang.srt.mem_cache.allocator import KVAllocator

    al = KVAllocator(size=40 * 64, device="cpu", max_batch_size=4, max_context_len=512, page_size=64)
    g = torch.Generator().manual_seed(3)
    al.free_slots = al.free_slots[torch.randperm(len(al.free_slots), generator=g)]  # scattered pages
    free0 = al.free_slots.clone()
    steps = [(0, 130, 0), (1, 64, 0), (2, 1, 0), (0, 1, 130), (1, 1, 64), (2, 63, 1), (2, 1, 64), (0, 70, 131), (3, 65, 0)]
    locs = []
    for req, need, alloced in steps:
        locs.append(al.alloc(req, need, alloced).numpy().astype(np.int32))
    np.savez_compressed(os.path.join(OUT, "kv_alloc.npz"), free_slots=free0.numpy(), steps=np.array(steps, np.int32),
                        req_to_page=al.req_to_page.numpy(), **{f"loc{i}": l for i, l in enumerate(locs)})
    print("alloc", al.req_to_page[:, :4].tolist())


def gen_gemm():
    ns = {"torch": torch}
    ref_test = "/root/reference/python/sglang/test/test_block_fp8.py"
    # sglang.srt.layers.activation does not import under the stubs (server_args -> openai pydantic models);
    # run the reference's own SiluAndMul.forward_native body straight from its source file instead.
    fwd = _ref_import.load_method_from_source("/root/reference/python/sglang/srt/layers/activation.py",
                                              "SiluAndMul", "forward_native",
                                              {"torch": torch, "F": torch.nn.functional})

    class SiluAndMul:  # thin holder so that torch_w8a8_block_fp8_moe's `SiluAndMul().forward_native(x)` resolves
        def forward_native(self, x):
            return fwd(self, x)

    ns["SiluAndMul"] = SiluAndMul
    _ref_import.load_functions_from_source(
        ref_test, {"native_per_token_group_quant_fp8", "native_w8a8_block_fp8_matmul", "torch_w8a8_block_fp8_moe"}, ns)
    quant, matmul, moe = ns["native_per_token_group_quant_fp8"], ns["native_w8a8_block_fp8_matmul"], ns["torch_w8a8_block_fp8_moe"]
    g = torch.Generator().manual_seed(20)
    fp8_max = 448.0
    # quant
    x = (torch.randn(83, 512, generator=g) * 3).to(torch.bfloat16)
    x[0] = 0
    x[1] *= 1e-12
    xq, xs = quant(x, 128)
    # block matmul (recipe of test_block_fp8.py:157-176)
    M, N, K = 37, 384, 512
    A = ((torch.rand(M, K, generator=g) - 0.5) * 2 * fp8_max).clamp(-fp8_max, fp8_max).to(torch.float8_e4m3fn)
    B = ((torch.rand(N, K, generator=g) - 0.5) * 2 * fp8_max).clamp(-fp8_max, fp8_max).to(torch.float8_e4m3fn)
    As = torch.rand(M, K // 128, generator=g) * 1e-2
    Bs = torch.rand(N // 128, K // 128, generator=g) * 1e-2
    C = matmul(A, B, As, Bs, [128, 128], output_dtype=torch.bfloat16)
    # MoE (recipe of test_block_fp8.py:268-289), E=4, topk=2
    Bm, D, I, E, topk = 9, 256, 128, 4, 2
    a = (torch.randn(Bm, D, generator=g) / 10).to(torch.bfloat16)
    w1 = ((torch.rand(E, 2 * I, D, generator=g) - 0.5) * 2 * fp8_max).clamp(-fp8_max, fp8_max).to(torch.float8_e4m3fn)
    w2 = ((torch.rand(E, D, I, generator=g) - 0.5) * 2 * fp8_max).clamp(-fp8_max, fp8_max).to(torch.float8_e4m3fn)
    w1_s = torch.rand(E, 2 * I // 128, D // 128, generator=g) * 1e-2
    w2_s = torch.rand(E, D // 128, I // 128, generator=g) * 1e-2
    score = torch.randn(Bm, E, generator=g).to(torch.bfloat16)
    moe_out = moe(a, w1, w2, w1_s, w2_s, score, topk, [128, 128])
    sw = torch.softmax(score, dim=-1, dtype=torch.float32)
    tw, ti = torch.topk(sw, topk)
    # silu
    y = (torch.randn(11, 256, generator=g) * 2).to(torch.bfloat16)
    act = SiluAndMul().forward_native(y)
    np.savez_compressed(os.path.join(OUT, "gemm_block_fp8.npz"),
                        quant_x=bf(x), quant_q=f8(xq), quant_s=xs.numpy(),
                        mm_A=f8(A), mm_B=f8(B), mm_As=As.numpy(), mm_Bs=Bs.numpy(), mm_C=bf(C),
                        moe_a=bf(a), moe_w1=f8(w1), moe_w2=f8(w2), moe_w1_s=w1_s.numpy(), moe_w2_s=w2_s.numpy(),
                        moe_topk_w=tw.numpy(), moe_topk_ids=ti.numpy().astype(np.int32), moe_out=bf(moe_out),
                        silu_x=bf(y), silu_out=bf(act))
    print("gemm", tuple(C.shape), tuple(moe_out.shape))


def gen_rmsnorm():
    """RMSNorm.forward_native run from the reference's own source (layernorm.py:88-112)."""
    fwd = _ref_import.load_method_from_source("/root/reference/python/sglang/srt/layers/layernorm.py", "RMSNorm",
                                              "forward_native",
                                              {"torch": torch, "Optional": __import__("typing").Optional,
                                               "Union": __import__("typing").Union, "Tuple": __import__("typing").Tuple})

    class Holder:
        pass

    g = torch.Generator().manual_seed(31)
    out = {}
    for name, T, H in (("h7168", 5, 7168), ("q1536", 7, 1536), ("kv512", 7, 512)):
        h = Holder()
        h.weight = (1.0 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16)
        h.variance_epsilon = 1e-6
        x = (torch.randn(T, H, generator=g) * 2).to(torch.bfloat16)
        r = torch.randn(T, H, generator=g).to(torch.bfloat16)
        y_plain = fwd(h, x)
        y_res, r_out = fwd(h, x, r)
        out.update({f"{name}_w": bf(h.weight), f"{name}_x": bf(x), f"{name}_r": bf(r), f"{name}_y": bf(y_plain),
                    f"{name}_y_res": bf(y_res), f"{name}_r_out": bf(r_out)})
    np.savez_compressed(os.path.join(OUT, "rmsnorm_native.npz"), **out)
    print("rmsnorm", sorted(out)[:3])


def gen_router():
    """biased_grouped_topk_impl (moe/topk.py:596-663), the torch statement behind `moe_fused_gate`, run eagerly from the
    reference's own source: DeepSeek-V3 routing (256 experts, 8 groups, top-4 groups, top-8, scaling 2.5) + smaller shapes."""
    import typing

    ns = {"torch": torch, "Optional": typing.Optional, "ExpertLocationDispatchInfo": object,
          "topk_ids_logical_to_physical": lambda ids, info: ids}
    path = "/root/reference/python/sglang/srt/layers/moe/topk.py"
    ns["_mask_topk_ids_padded_region"] = _ref_import.load_function_from_source(path, "_mask_topk_ids_padded_region", ns)
    impl = _ref_import.load_function_from_source(path, "biased_grouped_topk_impl", ns)
    g = torch.Generator().manual_seed(41)
    out = {}
    for name, T, E, G, TG, K, scale, on_out, npad in (("dsv3", 37, 256, 8, 4, 8, 2.5, True, None),
                                                       ("dsv3_noscale", 16, 256, 8, 4, 8, 2.5, False, 11),
                                                       ("e64", 9, 64, 4, 2, 6, 1.0, True, None),
                                                       ("e128_g1", 5, 128, 1, 1, 8, 1.5, True, None)):
        logits = torch.randn(T, E, generator=g) * 2
        bias = torch.randn(E, generator=g) * 0.1
        n = None if npad is None else torch.tensor(npad)
        w, ids = impl(torch.empty(T, 1), logits, bias, K, True, G, TG, 0, scale, n, None, on_out)
        out.update({f"{name}_logits": logits.numpy(), f"{name}_bias": bias.numpy(), f"{name}_w": w.numpy(),
                    f"{name}_ids": ids.numpy().astype(np.int32),
                    f"{name}_cfg": np.array([G, TG, K, int(on_out), -1 if npad is None else npad], np.int32),
                    f"{name}_scale": np.array([scale], np.float32)})
    np.savez_compressed(os.path.join(OUT, "router_biased_grouped_topk.npz"), **out)
    print("router", sorted(k for k in out if k.endswith("_ids")))


def gen_router_topk():
    """The plain top-k routers' torch statements run from the reference's own source: fused_topk_native (moe/topk.py:73-91; the
    statement behind flashinfer.topk_softmax, called by fused_topk :505-520) and fused_topk_bias (:51-70; the statement behind
    flashinfer.routing_flash, LongCat-Flash :836-845).  Both carry @torch.compile in the source: the decorator is a no-op here."""
    import typing
    import torch.nn.functional as F

    ns = {"torch": torch, "F": F, "Optional": typing.Optional, "ExpertLocationDispatchInfo": object,
          "topk_ids_logical_to_physical": lambda ids, info: ids, "get_compiler_backend": lambda: "eager"}
    path = "/root/reference/python/sglang/srt/layers/moe/topk.py"
    real_compile = torch.compile
    torch.compile = lambda *a, **k: (lambda f: f)
    try:
        native = _ref_import.load_function_from_source(path, "fused_topk_native", ns)
        biased = _ref_import.load_function_from_source(path, "fused_topk_bias", ns)
    finally:
        torch.compile = real_compile
    g = torch.Generator().manual_seed(43)
    out = {}
    for name, T, E, K, renorm in (("softmax_e128", 23, 128, 8, True), ("softmax_e8", 7, 8, 2, False), ("softmax_e256", 5, 256, 6, True),
                                  ("softmax_e96", 11, 96, 4, True)):
        logits = torch.randn(T, E, generator=g) * 2
        w, ids = native(torch.empty(T, 1), logits, K, renorm)
        out.update({f"{name}_logits": logits.numpy(), f"{name}_w": w.numpy(), f"{name}_ids": ids.numpy().astype(np.int32),
                    f"{name}_cfg": np.array([K, int(renorm)], np.int32)})
    for name, T, E, K, renorm in (("bias_longcat", 19, 768, 12, False), ("bias_e64", 9, 64, 6, True)):
        logits = torch.randn(T, E, generator=g) * 2
        bias = torch.randn(E, generator=g) * 0.01
        w, ids = biased(torch.empty(T, 1), logits, bias, K, renorm)
        out.update({f"{name}_logits": logits.numpy(), f"{name}_bias": bias.numpy(), f"{name}_w": w.numpy(),
                    f"{name}_ids": ids.numpy().astype(np.int32), f"{name}_cfg": np.array([K, int(renorm)], np.int32)})
    np.savez_compressed(os.path.join(OUT, "router_topk_plain.npz"), **out)
    print("router_topk", sorted(k for k in out if k.endswith("_ids")))


def gen_rope():
    """DeepseekScalingRotaryEmbedding (layers/rotary_embedding.py:719-846) run from the reference's own source: the YaRN
    cos/sin cache (fp32, as the CUDA path keeps it, :113-115) of a reduced config and forward_native on bf16 q_pe / k_pe —
    GPT-J style (DeepSeek-V3: is_neox_style=False, deepseek_v2.py:492-499) and NeoX style."""
    import math
    import typing

    path = "/root/reference/python/sglang/srt/layers/rotary_embedding.py"
    ns = {"torch": torch, "math": math, "Optional": typing.Optional, "Tuple": typing.Tuple, "Union": typing.Union}
    for fn in ("_rotate_neox", "_rotate_gptj", "_yarn_find_correction_dim", "_yarn_find_correction_range",
               "_yarn_linear_ramp_mask", "yarn_get_mscale"):
        ns[fn] = _ref_import.load_function_from_source(path, fn, ns)
    cls = "DeepseekScalingRotaryEmbedding"
    inv_freq = _ref_import.load_method_from_source(path, cls, "_compute_inv_freq", ns)
    cache_fn = _ref_import.load_method_from_source(path, cls, "_compute_cos_sin_cache", ns)
    fwd = _ref_import.load_method_from_source(path, cls, "forward_native", ns)

    class Rope:
        _compute_inv_freq = inv_freq

    g = torch.Generator().manual_seed(51)
    out = {}
    for name, neox, T, H in (("gptj", False, 19, 16), ("neox", True, 7, 5)):
        r = Rope()
        r.head_size = r.rotary_dim = 64
        r.base, r.max_position_embeddings, r.scaling_factor = 10000, 64, 4.0     # 256 cache rows (V3: 4096 x 40)
        r.extrapolation_factor, r.beta_fast, r.beta_slow, r.device, r.is_neox_style = 1, 32, 1, "cpu", neox
        r.mscale = float(ns["yarn_get_mscale"](4.0, 1.0) / ns["yarn_get_mscale"](4.0, 1.0) * 1.0)
        r.cos_sin_cache = cache_fn(r)
        pos = torch.randint(0, 256, (T,), generator=g)
        q = torch.randn(T, H, 64, generator=g).to(torch.bfloat16)
        k = torch.randn(T, 1, 64, generator=g).to(torch.bfloat16)
        if neox:   # forward_native's neox branch assumes [batch, seq] positions (:826-830)
            qo, ko = fwd(r, pos[None], q[None], k[None])
            qo, ko = qo[0], ko[0]
        else:
            qo, ko = fwd(r, pos, q, k)
        out.update({f"{name}_cache": r.cos_sin_cache.numpy(), f"{name}_pos": pos.numpy(), f"{name}_q": bf(q), f"{name}_k": bf(k),
                    f"{name}_q_out": bf(qo.contiguous()), f"{name}_k_out": bf(ko.contiguous())})
    np.savez_compressed(os.path.join(OUT, "rope_deepseek_yarn.npz"), **out)
    print("rope", {k: v.shape for k, v in out.items() if k.endswith("_out")})


def gen_kv_move():
    """move_kv_cache_native (mem_cache/memory_pool.py:2039-2052; the per_token_head MLA pool runs the same statement over its
    three buffers per layer, :746-763) from the reference's own source: 2 layers x (k_lora u8 [S,512], k_scale f32 [S,1],
    k_rope bf16 [S,64]) with OVERLAPPING source / target sets (a compaction: gather-all-then-scatter semantics)."""
    import typing

    ns = {"torch": torch, "List": typing.List}
    move = _ref_import.load_function_from_source("/root/reference/python/sglang/srt/mem_cache/memory_pool.py",
                                                 "move_kv_cache_native", ns)
    g = torch.Generator().manual_seed(61)
    S, L = 96, 2
    lora = [torch.randint(0, 256, (S, 512), generator=g, dtype=torch.uint8) for _ in range(L)]
    scale = [torch.rand(S, 1, generator=g) for _ in range(L)]
    rope = [torch.randn(S, 64, generator=g).to(torch.bfloat16) for _ in range(L)]
    src = torch.tensor([5, 6, 7, 8, 20, 21, 40, 41, 42, 43, 44, 90], dtype=torch.int64)
    tgt = torch.tensor([4, 5, 6, 7, 8, 9, 38, 39, 40, 41, 42, 0], dtype=torch.int64)     # 5..8, 40..42 are read AND written
    out = {"src": src.numpy(), "tgt": tgt.numpy()}
    for l in range(L):
        out.update({f"lora{l}": lora[l].numpy().copy(), f"scale{l}": scale[l].numpy().copy(), f"rope{l}": bf(rope[l]).copy()})
    move(lora + scale, rope + [torch.zeros(S, 1) for _ in range(L)], tgt, src)
    for l in range(L):
        out.update({f"lora{l}_out": lora[l].numpy(), f"scale{l}_out": scale[l].numpy(), f"rope{l}_out": bf(rope[l])})
    np.savez_compressed(os.path.join(OUT, "kv_move.npz"), **out)
    print("kv_move", S, L, len(src))


def gen_ep_scatter_gather():
    """ep_scatter / ep_gather of the DeepExecutor (moe/executors/deep_ep_executor.py:173-430): the reference's own Triton
    kernels run on the CPU by Triton's interpreter (run as: TRITON_INTERPRET=1 python oracle/gen_golden.py --only-ep).
    fp8 rows + 1x128 scales scattered into 128-aligned expert groups, bf16 expert outputs gathered back with top-k weights."""
    ns = _ref_import.load_triton_functions("/root/reference/python/sglang/srt/layers/moe/executors/deep_ep_executor.py",
                                           ("_fwd_kernel_ep_scatter_1", "_fwd_kernel_ep_scatter_2", "ep_scatter",
                                            "_fwd_kernel_ep_gather", "ep_gather"))
    g = torch.Generator().manual_seed(71)
    T, H, K, E = 45, 1024, 4, 4
    x = torch.randint(0, 255, (T, H), generator=g, dtype=torch.uint8)
    xs = torch.rand(T, H // 128, generator=g)
    topk = torch.stack([torch.randperm(E + 4, generator=g)[:K] for _ in range(T)]).to(torch.int64)
    topk[topk >= E] = -1                                    # experts of other ranks
    topk[7] = -1
    cnt = torch.bincount(topk[topk >= 0], minlength=E)
    padded = ((cnt + 127) // 128 * 128).to(torch.int32)     # "token num of per expert is aligned to 128" (:281)
    M = int(padded.sum())
    start = torch.zeros(E, dtype=torch.int32)
    out = torch.zeros(M, H, dtype=torch.uint8)
    outs = torch.zeros(M, H // 128)
    m_idx = torch.full((M,), -1, dtype=torch.int32)
    oidx = torch.full((T, K), -1, dtype=torch.int32)
    ns["ep_scatter"](x, xs, topk, padded, start, out, outs, m_idx, oidx)
    y = torch.zeros(M, H, dtype=torch.bfloat16)            # (padding rows stay zero: the fixture compresses)
    used = oidx[oidx >= 0].long()
    y[used] = torch.randn(used.numel(), H, generator=g).to(torch.bfloat16)
    w = torch.rand(T, K, generator=g)
    gathered = torch.zeros(T, H, dtype=torch.float32)     # fp32 output: the interpreter's f32->bf16 cast truncates, GPUs round
    ns["ep_gather"](y, topk, w, oidx, gathered)
    np.savez_compressed(os.path.join(OUT, "ep_scatter_gather.npz"), x=x.numpy(), xs=xs.numpy(), topk=topk.numpy(),
                        padded=padded.numpy(), start_after=start.numpy(), out=out.numpy(), outs=outs.numpy(),
                        m_idx=m_idx.numpy(), oidx=oidx.numpy(), y=bf(y), w=w.numpy(), gathered=gathered.numpy())
    print("ep_scatter_gather", T, H, K, E, M, start.tolist())


def gen_call_trace():
    """Call-trace fixture of the boundary: FlashMLABackend.forward_decode and the verify branch of forward_extend
    (flashmla_backend.py:88-256) are exec'd FROM THE REFERENCE SOURCE (their module does not import under stubs) against
    recording fakes of `flash_mla_fp8` / `flash_mla_swap`; what they pass — function, keyword names, shapes, dtypes,
    strides, scalars, and the view taken of the result — is written to tests/golden/flashmla_backend_call_trace.json.
    The GPU test replays every recorded call through the real modules."""
    import json
    import types

    from sglang.srt.model_executor.forward_batch_info import ForwardMode

    path = os.path.join(_ref_import.REF_ROOT, "sglang/srt/layers/attention/flashmla_backend.py")
    calls = []

    def describe(v):
        if torch.is_tensor(v):
            return {"tensor": True, "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
                    "stride": list(v.stride()), "contiguous": bool(v.is_contiguous())}
        if isinstance(v, (bool, int, float)) or v is None:
            return {"value": v}
        return {"repr": repr(v)}

    class Recorder(types.SimpleNamespace):
        def __init__(self, name):
            super().__init__()
            self._name = name

        def quantize_ckv_per_token_head(self, q, kv_lora_rank):
            calls.append({"module": self._name, "fn": "quantize_ckv_per_token_head", "args": [describe(q), describe(kv_lora_rank)], "kwargs": {}})
            lead = q.shape[:-1]
            return (torch.zeros(lead + (kv_lora_rank,), dtype=torch.float8_e4m3fn), torch.ones(lead + (1,), dtype=torch.float32),
                    torch.zeros(lead + (q.shape[-1] - kv_lora_rank,), dtype=torch.bfloat16))

        def _attend(self, fn, q_like, kw):
            calls.append({"module": self._name, "fn": fn, "args": [], "kwargs": {k: describe(v) for k, v in kw.items()}})
            bs, s_q, h = q_like.shape[:3]
            return torch.zeros(bs, s_q, h, kw["head_dim_v"], dtype=torch.bfloat16), torch.zeros(bs, h, s_q, dtype=torch.float32)

        def flash_mla_ckv_fp8_per_token(self, **kw):
            return self._attend("flash_mla_ckv_fp8_per_token", kw["q_nope"], kw)

        def flash_mla_with_kvcache(self, **kw):
            return self._attend("flash_mla_with_kvcache", kw["q"], kw)

    fp8_mod, swap_mod = Recorder("flash_mla_fp8"), Recorder("flash_mla_swap")
    ns = {"torch": torch, "PAGE_SIZE": 64, "ForwardMode": ForwardMode, "flash_mla_fp8": fp8_mod, "flash_mla_swap": swap_mod,
          "RadixAttention": object, "ForwardBatch": object, "Optional": __import__("typing").Optional}
    _ref_import.load_functions_from_source(path, {"get_flash_mla_module"}, ns)
    module_fn = ns["get_flash_mla_module"]
    forward_decode = _ref_import.load_method_from_source(path, "FlashMLABackend", "forward_decode", ns)
    forward_extend = _ref_import.load_method_from_source(path, "FlashMLABackend", "forward_extend", ns)
    # (FlashMLABackend.get_flash_mla_module, :85-86, is `return get_flash_mla_module(M, self.cache_quant_method)`: bound below
    #  by hand — exec'd next to the module-level function of the same name it would shadow it)

    def run(tag, method, H, bs, s_q, quant, cache_dtype, draft_token_num, seqlen_incr, mode):
        slots, max_pages = 64 * 9, 5
        if quant == "per_token_head":
            k_cache = (torch.zeros(slots, 1, 512, dtype=torch.uint8), torch.ones(slots, 1, 1), torch.zeros(slots, 1, 64, dtype=torch.bfloat16))
        else:
            k_cache = torch.zeros(slots, 1, 576, dtype=cache_dtype)
        pool = types.SimpleNamespace(get_key_buffer=lambda layer_id: k_cache,
                                     set_kv_buffer=lambda *a, **k: calls.append({"module": "token_to_kv_pool", "fn": "set_kv_buffer", "args": [describe(x) for x in a[1:]], "kwargs": {}}))
        fb = types.SimpleNamespace(out_cache_loc=torch.arange(bs * s_q), batch_size=bs, token_to_kv_pool=pool,
                                   seq_lens=torch.full((bs,), 100, dtype=torch.int64), forward_mode=mode)
        layer = types.SimpleNamespace(layer_id=0, tp_q_head_num=H, head_dim=576, v_head_dim=512, scaling=192 ** -0.5)
        meta = types.SimpleNamespace(block_table=torch.zeros(bs + 3, max_pages, dtype=torch.int32),   # graph buffers are larger than bs
                                     flashmla_metadata=torch.zeros(256, 8, dtype=torch.int32), num_splits=torch.zeros(bs + 1, dtype=torch.int32))
        self = types.SimpleNamespace(num_q_heads=H, multi_step_seqlen_incr=seqlen_incr, cache_quant_method=quant, cache_dtype=cache_dtype,
                                     kv_lora_rank=512, qk_rope_head_dim=64, kv_cache_dim=576, forward_metadata=meta,
                                     draft_token_num=draft_token_num)
        self.get_flash_mla_module = lambda M: module_fn(M, self.cache_quant_method)
        q = torch.zeros(bs * s_q, H * 576, dtype=torch.bfloat16)
        k = torch.zeros(bs * s_q, 1, 576, dtype=torch.bfloat16)
        n0 = len(calls)
        out = method(self, q, k, k, layer, fb, True)
        return {"case": tag, "H": H, "bs": bs, "s_q": s_q, "quant_method": quant, "cache_dtype": str(cache_dtype).replace("torch.", ""),
                "calls": calls[n0:], "returned": describe(out)}

    DEC, VER = ForwardMode.DECODE, ForwardMode.TARGET_VERIFY
    trace = [run("decode per-token fp8, H=128", forward_decode, 128, 3, 1, "per_token_head", torch.float8_e4m3fn, 0, 0, DEC),
             run("decode per-token fp8, H=16 (TP8 shard), MTP draft step 2", forward_decode, 16, 2, 1, "per_token_head", torch.float8_e4m3fn, 4, 2, DEC),
             run("decode plain fp8 cache, H=128 -> flash_mla_fp8", forward_decode, 128, 2, 1, "none", torch.float8_e4m3fn, 0, 0, DEC),
             run("decode plain fp8 cache, H=16 -> flash_mla_swap", forward_decode, 16, 2, 1, "none", torch.float8_e4m3fn, 0, 0, DEC),
             run("decode bf16 cache, H=16 -> flash_mla_swap", forward_decode, 16, 2, 1, "none", torch.bfloat16, 0, 0, DEC),
             run("verify s_q=4 per-token fp8, H=128", forward_extend, 128, 2, 4, "per_token_head", torch.float8_e4m3fn, 4, 0, VER),
             run("verify s_q=4 bf16 cache, H=8 -> flash_mla_swap", forward_extend, 8, 2, 4, "none", torch.bfloat16, 4, 0, VER)]
    with open(os.path.join(OUT, "flashmla_backend_call_trace.json"), "w") as f:
        json.dump({"source": "python/sglang/srt/layers/attention/flashmla_backend.py:88-256 exec'd by oracle/gen_golden.py:gen_call_trace",
                   "trace": trace}, f, indent=1)
    print("call trace:", sum(len(t["calls"]) for t in trace), "calls in", len(trace), "cases")


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--only-call-trace" in sys.argv:
        gen_call_trace()
        sys.exit(0)
    if "--only-ep" in sys.argv:
        gen_ep_scatter_gather()
        sys.exit(0)
    if "--only-kv-move" in sys.argv:
        gen_kv_move()
        sys.exit(0)
    if "--only-rope" in sys.argv:
        gen_rope()
        sys.exit(0)
    if "--only-router" in sys.argv:
        gen_router()
        gen_router_topk()
        sys.exit(0)
    if "--only-rmsnorm" in sys.argv:
        gen_rmsnorm()
        sys.exit(0)
    gen_mla_torch_native()
    gen_kv_quant()
    gen_alloc()
    gen_gemm()
    gen_rmsnorm()
    gen_router()
    gen_router_topk()
    gen_rope()
    gen_kv_move()
    if os.environ.get("TRITON_INTERPRET") == "1":
        gen_ep_scatter_gather()
    print("golden written to", OUT)
    gen_call_trace()
