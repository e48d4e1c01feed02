"""GPU box: randomized parity sweep of K2-bf16 (flash_mla_swap.flash_mla_with_kvcache: both workgroup shapes, split requests, causal s_q > 1,
NaN outside the valid tokens) against the exact float64 oracle.  usage: python tools/stress_mla_bf16.py [cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import flash_mla_swap as fsw
from helpers import rel_mae
from oracle import mla_ref
import test_mla_gpu as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst = 0.0
for it in range(N):
    H = rng.choice([8, 16, 24, 32, 40, 64, 96, 128])
    s_q = rng.choice([1, 1, 2, 4])
    bs = rng.randint(1, 12)
    lens = [rng.choice([0, 1, 31, 32, 33, 63, 64, 65, rng.randint(1, 2500)]) for _ in range(bs)]
    if s_q > 1:
        lens = [max(L, s_q) for L in lens]
    q, kc, bt, seq, pages = T.make_bf16_576_case(lens, H, s_q, seed=2000 + it)
    meta, ns = fsw.get_mla_metadata(seq.to(dev), s_q * H, 1)
    o, lse = fsw.flash_mla_with_kvcache(q.to(dev), kc.to(dev).view(pages, 64, 1, 576), bt.to(dev), seq.to(dev), 512, meta, ns, T.SCALE, True)
    torch.cuda.synchronize()
    ref, rlse = mla_ref.mla_decode_with_kvcache(q, torch.nan_to_num(kc).view(pages, 64, 1, 576), bt, seq, 512, T.SCALE, True)
    o, lse = o.cpu(), lse.cpu()
    assert torch.isfinite(o.float()).all(), (it, "NaN/inf leaked")
    rel = rel_mae(o, ref)
    m = torch.isfinite(rlse)
    assert torch.equal(torch.isfinite(lse), m), it
    dl = float((lse[m].double() - rlse[m]).abs().max()) if m.any() else 0.0
    assert rel < 6e-3 and dl < 2e-3, (it, H, s_q, lens, rel, dl)
    worst = max(worst, rel)
    print(f"case {it}: H={H} s_q={s_q} bs={bs} lens={lens[:6]}{'...' if bs > 6 else ''} max_splits={int((ns[1:]-ns[:-1]).max())} rel-MAE {rel:.3e} lse {dl:.1e}")
print(f"all {N} cases within tolerance; worst rel-MAE {worst:.3e}")
