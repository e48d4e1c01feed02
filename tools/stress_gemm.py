"""GPU box: randomized parity sweep of the grouped GEMM dispatch (128-row tiles and the 256x256 tile) against the oracle."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import deep_gemm
import test_gemm_gpu as T
from oracle import gemm_ref
N_CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = T.DEV
worst = 0.0
for it in range(N_CASES):
    E = rng.randint(1, 9)
    big = rng.random() < 0.6
    counts = [rng.choice([0, 1, 31, 33, 128, 129, rng.randint(150, 700)]) if big else rng.choice([0, 1, 5, 32, 33, 70]) for _ in range(E)]
    N = rng.choice([128, 260, 384, 512, 640, 772])
    K = rng.choice([256, 512, 768, 1024])
    if sum(counts) == 0:
        counts[0] = 3
    xq, xs, W, Ws, ex = T.make_group_case(counts, N, K, seed=500 + it)
    M = xq.shape[0]
    out = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device=DEV)
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq.to(DEV), xs.to(DEV)), (W.to(DEV), Ws.to(DEV)), out[:M], ex.to(DEV), use_pdl=True)
    ref = gemm_ref.grouped_gemm_offset(xq, xs, W, Ws, ex)
    rel = T.rel_mae(out[:M].cpu(), ref)
    assert rel < 1e-3, (it, counts, N, K, rel)
    assert bool((out[M:] == 7.0).all())
    worst = max(worst, rel)
    print(f"case {it}: counts={counts} N={N} K={K} avg={M/E:.0f} rel-MAE {rel:.2e}")
print(f"all {N_CASES} cases within 1e-3; worst {worst:.2e}")
