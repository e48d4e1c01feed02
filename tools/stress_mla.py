"""GPU box: randomized parity sweep of the MLA decode dispatch (every mapping the shape selects) against the exact oracle."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import flash_mla_fp8 as fm
from helpers import make_paged_case
import test_mla_gpu as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for it in range(N):
    H = rng.choice([8, 16, 24, 40, 64, 96, 128])
    s_q = rng.choice([1, 1, 2, 4])
    bs = rng.randint(1, 12)
    lens = [rng.choice([0, 1, 63, 64, 65, rng.randint(1, 2500)]) for _ in range(bs)]
    if s_q > 1:
        lens = [max(L, s_q) for L in lens]
    c = make_paged_case(lens, H, s_q=s_q, seed=1000 + it)
    o, lse, ref, rlse, ns = T.run_decode(fm, c, H, s_q, emulate=False)
    rel = T.check(o, lse, ref, rlse, f"case{it}")
    worst = max(worst, rel)
    print(f"case {it}: H={H} s_q={s_q} bs={bs} lens={lens[:6]}{'...' if bs > 6 else ''} max_splits={int((ns[1:]-ns[:-1]).max())} rel-MAE {rel:.3e}")
print(f"all {N} cases within tolerance; worst rel-MAE {worst:.3e}")
