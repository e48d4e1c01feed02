"""GPU box: dense block-fp8 GEMM (G4, deep_gemm.gemm_fp8_fp8_bf16_nt) at the decode shapes of the MLA projections."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, deep_gemm
from fluent_mi355.gemm import per_token_group_quant_fp8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def run(T, N, K, reps=20, copies=8):
    Ws = [(torch.randint(0, 120, (N, K), device=dev, generator=g, dtype=torch.uint8).view(torch.float8_e4m3fn),
           torch.rand((N + 127) // 128, K // 128, device=dev, generator=g) * 1e-2) for _ in range(copies)]   # no cache reuse
    x = torch.randn(T, K, device=dev, generator=g).to(torch.bfloat16)
    xq, xs = per_token_group_quant_fp8(x, column_major_scales=True)
    out = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
    for w in Ws: deep_gemm.gemm_fp8_fp8_bf16_nt((xq, xs), w, out)
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        deep_gemm.gemm_fp8_fp8_bf16_nt((xq, xs), Ws[0], out)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(gr):
        for w in Ws: deep_gemm.gemm_fp8_fp8_bf16_nt((xq, xs), w, out)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * copies)
    print(json.dumps({"T": T, "N": N, "K": K, "us": round(us, 1), "weight_GBs": round(N * K / us / 1e3, 1), "hbm_frac": round(N * K / us / 1e3 / 8000, 3)}))
for T in (int(a) for a in (sys.argv[1:] or ["128", "256"])):
    run(T, 7168, 16384)
    run(T, 2176, 7168); run(T, 3072, 1536); run(T, 7168, 2048); run(T, 7168 * 2, 7168)
