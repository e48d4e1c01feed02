"""GPU box: random-shape stress of the round-3 bf16 kernels — B2 (absorption bmm's), B3 (router GEMM) against fp32 torch products and A2
(fused absorb + RoPE + K5 + K4) against the four-launch chain, bit for bit.  usage: python tools/stress_absorb.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import flash_mla_fp8 as fm, flashinfer
from fluent_mi355.bmm import bmm
from fluent_mi355.rope import apply_rope_with_cos_sin_cache_inplace
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
def ulp_ok(got, ref32):
    return bool(((got.float() - ref32.to(torch.bfloat16).float()).abs() <= ref32.abs() * 2.0 ** -7 + 2e-3).all())
worst = 0
for c in range(cases):
    T = int(torch.randint(1, 700, (1,), generator=g)) if c % 3 else int(torch.randint(1, 70, (1,), generator=g))
    H = [8, 16, 32, 128][int(torch.randint(0, 4, (1,), generator=g))]
    neox = bool(torch.randint(0, 2, (1,), generator=g))
    q = (torch.randn(T, H, 192, generator=g) * torch.exp(torch.randn(T, H, 1, generator=g))).to(torch.bfloat16).to(dev)
    w = (torch.randn(H, 512, 128, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    w_kc = w.transpose(1, 2)
    lat = (torch.randn(T, 576, generator=g) * 2).to(torch.bfloat16).to(dev)
    cache = torch.randn(512, 64, generator=g).to(dev)
    pos = torch.randint(-2, 520, (T,), generator=g).to(dev)          # (out-of-range positions are clamped by both paths)
    slots = 1024
    loc = torch.randint(-3, slots + 3, (T,), generator=g).to(torch.int32)
    loc = torch.where(torch.rand(T, generator=g) < 0.8, torch.randperm(slots * 2, generator=g)[:T].to(torch.int32) % slots, loc).to(dev)
    # duplicate cache rows would make the chain's and the fused launch's write ORDER matter: keep them unique where in range
    u = torch.unique(loc[(loc >= 0) & (loc < slots)])
    if u.numel() != int(((loc >= 0) & (loc < slots)).sum()):
        loc = (torch.randperm(slots, generator=g)[:T].to(torch.int32) if T <= slots else loc).to(dev)
    mk = lambda: [torch.zeros(slots, 1, 512, dtype=torch.uint8, device=dev), torch.zeros(slots, 1, 1, device=dev), torch.zeros(slots, 1, 64, dtype=torch.bfloat16, device=dev)]
    la, ca = lat.clone(), mk()
    Q = torch.empty(T, H, 576, dtype=torch.bfloat16, device=dev)
    bmm(q[..., :128].transpose(0, 1), w_kc, out=Q[..., :512].transpose(0, 1))
    assert ulp_ok(Q[..., :512], torch.einsum("thk,hkn->thn", q[..., :128].float(), w_kc.float())), ("q-absorb", T, H)
    K = la.unsqueeze(1)
    apply_rope_with_cos_sin_cache_inplace(pos, q[..., 128:], K[..., 512:], 64, cache, is_neox=neox, output_q_rope=Q[..., 512:])
    rn, rs, rr = fm.quantize_q_and_cache_k(Q, K.contiguous(), ca[0], ca[1], ca[2], loc, 512)
    lb, cb = lat.clone(), mk()
    qn, qs, qr = fm.absorb_rope_quant(q, w_kc, pos, cache, lb, cb[0], cb[1], cb[2], loc, is_neox=neox)
    torch.cuda.synchronize()
    ok = (torch.equal(la.view(torch.int16), lb.view(torch.int16)) and torch.equal(qs, rs) and torch.equal(qn.view(torch.uint8), rn.view(torch.uint8))
          and torch.equal(qr.view(torch.int16), rr.view(torch.int16)) and all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(ca, cb)))
    assert ok, ("fused vs chain", T, H, neox)
    att = torch.randn(T, H, 512, generator=g).to(torch.bfloat16).to(dev)
    w_vc = (torch.randn(H, 128, 512, generator=g) * 0.05).to(torch.bfloat16).to(dev).transpose(1, 2)
    assert ulp_ok(bmm(att.transpose(0, 1), w_vc), torch.einsum("thk,hkn->htn", att.float(), w_vc.float())), ("v-absorb", T, H)
    x = torch.randn(T, 7168, generator=g).to(torch.bfloat16).to(dev)
    wr = (torch.randn(256, 7168, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    lg = flashinfer.dsv3_router_gemm(x, wr, out_dtype=torch.float32)
    ref = x.float() @ wr.float().t()
    err = float((lg - ref).abs().max()) / (float(ref.abs().max()) + 1e-9)
    worst = max(worst, err)
    assert err <= 1e-4, ("router", T, err)
    lg2 = flashinfer.dsv3_router_gemm(x, wr, out_dtype=torch.float32)
    assert torch.equal(lg, lg2), ("router determinism", T)
print(f"all {cases} cases: absorption bmm's within one bf16 ulp, fused launch bit-identical to the chain, router GEMM rel err <= {worst:.2e} and run-to-run identical")
