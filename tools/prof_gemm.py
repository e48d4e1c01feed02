"""GPU box: a SMALL compute-regime grouped-GEMM workload for profiler passes (rocprofv3 --pmc serialises kernels and the full
bench_gemm.py set-up — 11 GB of random weights per pass — takes minutes): 32 experts x 512 rows, w13 [4096, 7168] and w2
[7168, 2048], random fp8 bytes, 4 launches each.  The tiles, k loops and tile counts per CU round are those of BASELINE config 3
at T = 16384 (512 rows per expert); only the number of experts is smaller.  usage: tools/prof_gemm.py [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import deep_gemm
dev = torch.device("cuda:0")
E, R = 32, 512
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator(device=dev).manual_seed(0)
def rnd8(*shape):
    b = torch.randint(0, 255, shape, device=dev, generator=g, dtype=torch.int16)
    return torch.where((b & 0x7F) == 0x7F, b - 1, b).to(torch.uint8).view(torch.float8_e4m3fn)
M = E * R
ex = (torch.arange(E + 1, device=dev) * R).to(torch.int32)
for (N, K) in ((4096, 7168), (7168, 2048)):
    W, Ws = rnd8(E, N, K), torch.rand(E, N // 128, K // 128, device=dev, generator=g) * 1e-2
    A, As = rnd8(M, K), torch.rand(M, K // 128, device=dev, generator=g) * 1e-2 + 1e-3
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(n):
        deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True); e1.record()
    torch.cuda.synchronize()
    print(f"N={N} K={K} M={M}: {e0.elapsed_time(e1):.3f} ms = {2.0 * M * N * K / e0.elapsed_time(e1) / 1e9:.0f} TFLOP/s")
