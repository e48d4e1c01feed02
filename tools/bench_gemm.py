"""GPU box: FP8 block-scaled grouped GEMM (BASELINE config 3: DeepSeek-V3 MoE, 256 experts top-8, hidden 7168, inter 2048,
TP=1) in the decode regime (HBM-bound on weights) and the compute regime (fp8 MFMA-bound). Prints one JSON per T."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import deep_gemm, flashinfer
from eps.executor import silu

dev = torch.device("cuda:0")
HID, INTER, E, TOPK = 7168, 2048, int(os.environ.get("E", 256)), 8
MFMA_PEAK_TF, HBM_PEAK = 5000.0, 8000.0
g = torch.Generator(device=dev).manual_seed(0)

def fp8_rand(*shape):
    out = torch.empty(shape, dtype=torch.float8_e4m3fn, device=dev)
    flat = out.view(-1).view(torch.uint8)
    step = 1 << 28
    for i in range(0, flat.numel(), step):   # random bytes avoiding NaN patterns 0x7f/0xff
        n = min(step, flat.numel() - i)
        b = torch.randint(0, 255, (n,), device=dev, generator=g, dtype=torch.int16)
        b = torch.where((b & 0x7F) == 0x7F, b - 1, b)
        flat[i:i + n] = b.to(torch.uint8)
    return out

w13 = fp8_rand(E, 2 * INTER, HID)
w2 = fp8_rand(E, HID, INTER)
w13s = torch.rand(E, 2 * INTER // 128, HID // 128, device=dev, generator=g) * 1e-2
w2s = torch.rand(E, HID // 128, INTER // 128, device=dev, generator=g) * 1e-2

def run(T, iters):
    M = T * TOPK
    ids = torch.stack([torch.randperm(E, device=dev, generator=g)[:TOPK] for _ in range(min(T, 4096))])
    ids = ids.repeat((T + ids.shape[0] - 1) // ids.shape[0], 1)[:T].reshape(-1)
    counts = torch.bincount(ids, minlength=E)
    ex = torch.zeros(E + 1, dtype=torch.int32, device=dev); ex[1:] = torch.cumsum(counts, 0)
    hit = int((counts > 0).sum())
    x = (torch.randn(M, HID, device=dev, generator=g) / 10).to(torch.bfloat16)
    mp = (M + E * 31) // 32 * 32
    xq = torch.empty(M, HID, dtype=torch.float8_e4m3fn, device=dev)
    xs = torch.empty((HID // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2)
    gate_up = torch.empty(M, 2 * INTER, dtype=torch.bfloat16, device=dev)
    dq = torch.empty(M, INTER, dtype=torch.float8_e4m3fn, device=dev)
    ds = torch.empty((INTER // 128, mp), dtype=torch.float32, device=dev).permute(-1, -2)
    out = torch.empty(M, HID, dtype=torch.bfloat16, device=dev)
    def layer():
        flashinfer.quantization.quant_1x128(x, xq, xs, ex, E, (M + 3) // 4 * 4, mp, HID)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq, xs), (w13, w13s), gate_up, ex, use_pdl=True)
        a = silu(gate_up, ex, M)
        flashinfer.quantization.quant_1x128(a, dq, ds, ex, E, (M + 3) // 4 * 4, mp, INTER)
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((dq, ds), (w2, w2s), out, ex, use_pdl=True)
    def g1():
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq, xs), (w13, w13s), gate_up, ex, use_pdl=True)
    def g2():
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((dq, ds), (w2, w2s), out, ex, use_pdl=True)
    def timeit(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3
    layer()
    t1, t2, tl = timeit(g1), timeit(g2), timeit(layer)
    f1, f2 = 2.0 * M * 2 * INTER * HID, 2.0 * M * HID * INTER
    b1 = hit * (2 * INTER * HID) + M * (HID + HID // 128 * 4) + M * 2 * INTER * 2
    b2 = hit * (HID * INTER) + M * (INTER + INTER // 128 * 4) + M * HID * 2
    print(json.dumps({"T": T, "rows": M, "experts_hit": hit, "rows_per_expert": round(M / E, 1),
        "gate_up": {"ms": round(t1 * 1e3, 3), "TFLOPs": round(f1 / t1 / 1e12, 1), "mfma_frac": round(f1 / t1 / 1e12 / MFMA_PEAK_TF, 4), "GBs": round(b1 / t1 / 1e9, 1), "hbm_frac": round(b1 / t1 / 1e9 / HBM_PEAK, 4)},
        "down": {"ms": round(t2 * 1e3, 3), "TFLOPs": round(f2 / t2 / 1e12, 1), "mfma_frac": round(f2 / t2 / 1e12 / MFMA_PEAK_TF, 4), "GBs": round(b2 / t2 / 1e9, 1), "hbm_frac": round(b2 / t2 / 1e9 / HBM_PEAK, 4)},
        "moe_layer_ms(quant+gemm+silu+quant+gemm)": round(tl * 1e3, 3)}))

for T, it in ((128, 20), (256, 20), (512, 16), (1024, 10), (2048, 8), (4096, 6), (6144, 5), (8192, 4), (16384, 3)):
    if (len(sys.argv) > 1 and str(T) not in sys.argv[1:]) or (len(sys.argv) == 1 and T not in (128, 1024, 16384)):
        continue
    run(T, it)
