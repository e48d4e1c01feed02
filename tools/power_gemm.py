"""GPU box: grouped GEMM w13 (T = 16384 tokens top-8 over 256 experts, random fp8 bytes) run back to back for a few seconds under bench.py's
10 Hz clock / power sampler: TFLOP/s, mean shader clock, mean socket power -> time per launch AND cycles per launch (on a power-capped chip the
two differ between variants).  usage: FLUENT_MI355_LIB=... python tools/power_gemm.py [seconds]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import bench, deep_gemm
dev = torch.device("cuda:0")
HID, N, E, TOPK, T = 7168, 4096, 256, 8, 16384
g = torch.Generator(device=dev).manual_seed(0)
w = torch.empty(E, N, HID, dtype=torch.float8_e4m3fn, device=dev)
flat = w.view(-1).view(torch.uint8)
for i in range(0, flat.numel(), 1 << 28):
    n = min(1 << 28, flat.numel() - i)
    b = torch.randint(0, 255, (n,), device=dev, generator=g, dtype=torch.int16)
    flat[i:i + n] = torch.where((b & 0x7F) == 0x7F, b - 1, b).to(torch.uint8)
ws = torch.rand(E, N // 128, HID // 128, device=dev, generator=g) * 1e-2
M = T * TOPK
ids = torch.stack([torch.randperm(E, device=dev, generator=g)[:TOPK] for _ in range(2048)]).repeat(T // 2048, 1).reshape(-1)
counts = torch.bincount(ids, minlength=E)
ex = torch.zeros(E + 1, dtype=torch.int32, device=dev); ex[1:] = torch.cumsum(counts, 0)
xq = (torch.randn(M, HID, device=dev, generator=g) / 10).to(torch.float8_e4m3fn)
xs = torch.rand(M, HID // 128, device=dev, generator=g) * 1e-2 + 1e-3
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
run = lambda: deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq, xs), (w, ws), out, ex, use_pdl=True)
s, smp = bench.sampled_loop(run, float(sys.argv[1]) if len(sys.argv) > 1 else 3.0, 8)
tf = 2.0 * M * N * HID / s / 1e12
print(json.dumps({"lib": os.path.basename(os.environ.get("FLUENT_MI355_LIB", "product")), "big": os.environ.get("FLUENT_GEMM_BIG", ""), "ms": round(s * 1e3, 3),
                  "TFLOPs": round(tf, 1), "sclk_mhz": smp["sclk_mhz_mean"], "power_w": smp["power_w_mean"], "cap_w": smp["power_cap_w"],
                  "Mcycles_per_launch": round(s * (smp["sclk_mhz_mean"] or 0) , 1), "samples": smp["samples"], "src": smp["source"]}))
