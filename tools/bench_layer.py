"""GPU box: ONE rank's share of a DeepSeek-V3 decoder layer at BASELINE config 4 (TP8 attention + EP8 MoE, bs=256, seq=8192)
— the hot-path op sequence on synthetic weights, captured in one hipGraph per layer group:
  fused add+RMSNorm+1x128 quant (C5 at world 1) -> q_a/kv_a projection (dense fp8 GEMM) -> K5 store, K4 quantise q, K1 MLA
  decode over the rank's 16 heads (latent KV replicated: every rank streams all 256 requests) -> o_proj (quant + dense fp8
  GEMM) -> fused add+RMSNorm+quant (C6 at world 1) -> routed rows of this rank's 32 experts: quant_1x128 -> grouped GEMM w13
  -> SiLU*mul -> quant_1x128 -> grouped GEMM w2.
The weight-absorption bmm / RoPE / router of the reference are not part of this repo: q is synthetic.
Prints one JSON line: ms per layer (whole, attention part, MoE part) and the bytes each part has to stream."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_fp8 as fm
import deep_gemm, flashinfer
import flashinfer.comm as comm
from eps.executor import silu
from eps.fast_ep import AllToAll
from fluent_mi355.gemm import per_token_group_quant_fp8

dev = torch.device("cuda:0")
DP_ATTN = os.environ.get("ATTN", "tp") == "dp"   # dp: DP-attention (this rank owns bs/8 requests with all 128 heads, KV partitioned)
BS, SEQ, H, HID, INTER, EL, TOPK, WORLD = 256, int(os.environ.get("SEQ", 8192)), (128 if DP_ATTN else 16), 7168, 2048, 32, 8, 8
BS_ATTN = BS // WORLD if DP_ATTN else BS
LAYERS = int(os.environ.get("LAYERS", 2))
g = torch.Generator(device=dev).manual_seed(0)
def fp8w(*shape):
    b = torch.randint(0, 255, shape, device=dev, generator=g, dtype=torch.int16)
    return torch.where((b & 0x7F) == 0x7F, b - 1, b).to(torch.uint8).view(torch.float8_e4m3fn)
def ws(*shape):
    return torch.rand(*shape, device=dev, generator=g) * 1e-2

wl = bench.build_workload(dev, LAYERS, BS_ATTN, SEQ, H, seed=3)
pages = wl["pages"]
meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
_, wsp = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(0, 1, BS, HID)
W = []
for l in range(LAYERS):
    W.append(dict(
        qkv_a=(fp8w(2112 + 128 - 2112 % 128 if 2112 % 128 else 2112, HID), None), gamma1=torch.ones(HID, dtype=torch.bfloat16, device=dev),
        gamma2=torch.ones(HID, dtype=torch.bfloat16, device=dev),
        o=(fp8w(HID, 16 * 128), ws(HID // 128, 16 * 128 // 128)),
        w13=(fp8w(EL, 2 * INTER, HID), ws(EL, 2 * INTER // 128, HID // 128)),
        w2=(fp8w(EL, HID, INTER), ws(EL, HID // 128, INTER // 128))))
    n_a = W[-1]["qkv_a"][0].shape[0]
    W[-1]["qkv_a"] = (W[-1]["qkv_a"][0], ws(n_a // 128, HID // 128))
x = torch.randn(BS, HID, device=dev, generator=g).to(torch.bfloat16)
res = torch.randn(BS, HID, device=dev, generator=g).to(torch.bfloat16)
attn_o = torch.randn(BS, 16 * 128, device=dev, generator=g).to(torch.bfloat16)     # stands in for the absorbed-V output
# EP dispatch / combine at world 1 (the exchange is a copy; the route / sort / gather / scatter / weighted-combine kernels
# run as in the 8-rank job): this rank's BS/WORLD tokens, top-8 over the 32 local experts
T_LOC = BS // WORLD
EP_OPS = os.environ.get("EP_OPS", "1") == "1"   # the device side of eps.fast_ep dispatch/combine (world 1: no exchange)
a2a = AllToAll(TOPK, EL, HID, T_LOC, None)
dp_x = torch.randn(T_LOC, HID, device=dev, generator=g).to(torch.bfloat16)
route = torch.stack([torch.randperm(EL, device=dev, generator=g)[:TOPK] for _ in range(T_LOC)]).to(torch.int32)
route_w = torch.rand(T_LOC, TOPK, device=dev, generator=g)
a2a_ex = torch.empty(EL + 1, dtype=torch.int32, device=dev)
a2a_rows = torch.zeros(T_LOC * TOPK, HID, dtype=torch.bfloat16, device=dev)
a2a_out = torch.empty(T_LOC, HID, dtype=torch.bfloat16, device=dev)
# routed rows of this rank: BS*TOPK/WORLD rows spread over its 32 experts
M = BS * TOPK // WORLD
counts = torch.bincount(torch.randint(0, EL, (M,), device=dev, generator=g), minlength=EL)
ex = torch.zeros(EL + 1, dtype=torch.int32, device=dev); ex[1:] = torch.cumsum(counts, 0)
mp = (M + EL * 31) // 32 * 32
rows = torch.randn(M, HID, device=dev, generator=g).to(torch.bfloat16)
buf = dict(norm=torch.empty_like(x), res=torch.empty_like(x), q8=torch.empty(BS, HID, dtype=torch.float8_e4m3fn, device=dev),
           s8=torch.empty(HID // 128, (BS + 3) // 4 * 4, dtype=torch.float32, device=dev).permute(-1, -2)[:BS],
           qkv=torch.empty(BS, W[0]["qkv_a"][0].shape[0], dtype=torch.bfloat16, device=dev),
           o=torch.empty(BS, HID, dtype=torch.bfloat16, device=dev),
           xq=torch.empty(M, HID, dtype=torch.float8_e4m3fn, device=dev),
           xs=torch.empty((HID // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2),
           gate_up=torch.empty(M, 2 * INTER, dtype=torch.bfloat16, device=dev),
           dq=torch.empty(M, INTER, dtype=torch.float8_e4m3fn, device=dev),
           ds=torch.empty((INTER // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2),
           down=torch.empty(M, HID, dtype=torch.bfloat16, device=dev))

def norm_quant(l, gamma):
    comm.trtllm_allreduce_fusion(allreduce_in=x, world_size=1, world_rank=0, token_num=BS, hidden_dim=HID, workspace_ptrs=wsp,
                                 pattern_code=comm.AllReduceFusionPattern.kARResidualRMSNormFP8BlockWiseQuant, residual_in=res,
                                 residual_out=buf["res"], norm_out=buf["norm"], quant_out=buf["q8"], scale_out=buf["s8"],
                                 rms_gamma=gamma, rms_eps=1e-6)
def attention(l):
    norm_quant(l, W[l]["gamma1"])
    deep_gemm.gemm_fp8_fp8_bf16_nt((buf["q8"], buf["s8"]), W[l]["qkv_a"], buf["qkv"])
    bench.layer_call(fm, wl, l, meta, ns)                                 # K5 + K4 + K1 (H=16, all 256 requests)
    oq, os_ = per_token_group_quant_fp8(attn_o, column_major_scales=True)
    deep_gemm.gemm_fp8_fp8_bf16_nt((oq, os_), W[l]["o"], buf["o"])
def moe(l):
    norm_quant(l, W[l]["gamma2"])
    if EP_OPS: a2a.dispatch(out_exclusive_sum=a2a_ex, out_expert_x=a2a_rows, dp_x=dp_x, indices=route, num_global_tokens=T_LOC)
    flashinfer.quantization.quant_1x128(rows, buf["xq"], buf["xs"], ex, EL, (M + 3) // 4 * 4, mp, HID)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((buf["xq"], buf["xs"]), W[l]["w13"], buf["gate_up"], ex, use_pdl=True)
    a = silu(buf["gate_up"], ex, M)
    flashinfer.quantization.quant_1x128(a, buf["dq"], buf["ds"], ex, EL, (M + 3) // 4 * 4, mp, INTER)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((buf["dq"], buf["ds"]), W[l]["w2"], buf["down"], ex, use_pdl=True)
    if EP_OPS: a2a.combine(out_tokens=a2a_out, weights=route_w, expert_y=buf["down"], num_global_tokens=T_LOC)

def timed(fn):
    for l in range(LAYERS): fn(l)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for l in range(LAYERS): fn(l)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for l in range(LAYERS): fn(l)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * LAYERS)

t_attn, t_moe = timed(attention), timed(moe)
t_all = timed(lambda l: (attention(l), moe(l)))
kv_bytes = bench.algorithmic_bytes(BS_ATTN, SEQ, H, 1)
w_attn = sum(W[0][k][0].numel() for k in ("qkv_a", "o"))
w_moe = int((counts > 0).sum()) * (2 * INTER * HID + HID * INTER)
print(json.dumps({"workload": f"one rank of {'DP8-attention' if DP_ATTN else 'TP8-attention'}/EP8 DeepSeek-V3 decoder layer, bs={BS} seq={SEQ}: attention H={H} over {BS_ATTN} requests + 32 local experts, {M} routed rows",
                  "ms_per_layer": round(t_all, 4), "attention_ms": round(t_attn, 4), "moe_ms": round(t_moe, 4),
                  "attention_GB": round((kv_bytes + w_attn) / 1e9, 3), "moe_weight_GB": round(w_moe / 1e9, 3),
                  "GBs_whole_layer": round((kv_bytes + w_attn + w_moe) / t_all / 1e6, 1),
                  "decode_tokens_per_s_8gpu_job(61 layers)": round(BS / (t_all * 61) * 1e3, 1)}))
