"""GPU box: the ragged cfg2 workload (128 requests, lengths 2048..6144, H=128: ~122 requests split in two and merged INSIDE
the decode kernel by their first piece, which waits for the arrival counts of the others) launched many times on the same metadata; every output is compared bit for
bit with the first.  A merge that read a partial row too early, a counter that was not put back to zero, or a lost
arrival would show as a mismatch (or a hang: run under `timeout`).  usage: tools/determinism_ragged.py [iterations]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_fp8 as fm
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
BS, H = 128, 128
g = torch.Generator(device=dev).manual_seed(5)
lens = torch.randint(2048, 6145, (BS,), device=dev, generator=g, dtype=torch.int32)
npg = ((lens + 63) // 64).tolist()
mp, pages = max(npg), sum(npg) + 1
slots = pages * 64
key = torch.randn(slots, 1, 576, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
k_lora = torch.empty(slots, 1, 512, dtype=torch.uint8, device=dev)
k_scale = torch.empty(slots, 1, 1, dtype=torch.float32, device=dev)
k_rope = torch.empty(slots, 1, 64, dtype=torch.bfloat16, device=dev)
fm.quantize_and_cache_k(key, k_lora, k_scale, k_rope, torch.arange(slots, dtype=torch.int32, device=dev), 512)
del key
perm = (torch.randperm(pages - 1, device=dev, generator=g).to(torch.int32) + 1).cpu()
bt = torch.zeros(BS, mp, dtype=torch.int32)
o = 0
for b, n in enumerate(npg):
    bt[b, :n] = perm[o:o + n]; o += n
bt = bt.to(dev)
q = torch.randn(BS, 1, H, 576, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
qn, qs, qr = fm.quantize_ckv_per_token_head(q, 512)
meta, ns = fm.get_mla_metadata(lens, H, 1)
def run():
    return fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                          k_scale.view(pages, 64, 1, 1), bt, lens, 512, meta, ns, bench.SCALE, True)
o0, l0 = run(); o0, l0 = o0.clone(), l0.clone()
bad = 0
side = torch.cuda.Stream()
# BESIDE_GEMM=1: a compute-regime grouped GEMM (256 x 256 tiles, one 8-wave workgroup per CU, ~5 ms) runs on the second stream the whole time, so
# that the decode kernel's workgroups are dispatched late, out of step and onto whatever CUs come free: the merging piece of a split request
# then really waits for pieces whose workgroups have not started (forward progress of the in-kernel split merge under CU contention)
gemm = None
if os.environ.get("BESIDE_GEMM") == "1":
    import deep_gemm
    E, R, Nn, K = 32, 2048, 4096, 7168
    W = torch.randint(0, 120, (E, Nn, K), device=dev, generator=g, dtype=torch.uint8).view(torch.float8_e4m3fn)
    Ws = torch.rand(E, Nn // 128, K // 128, device=dev, generator=g) * 1e-2
    A = torch.randint(0, 120, (E * R, K), device=dev, generator=g, dtype=torch.uint8).view(torch.float8_e4m3fn)
    As = torch.rand(E * R, K // 128, device=dev, generator=g)
    ex = (torch.arange(E + 1, device=dev) * R).to(torch.int32)
    gout = torch.empty(E * R, Nn, dtype=torch.bfloat16, device=dev)
    gemm = lambda: deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), gout, ex, use_pdl=True)
    gemm(); torch.cuda.synchronize()
for i in range(N):
    if gemm is not None:
        if i % 8 == 0:
            with torch.cuda.stream(side):
                gemm()
    elif i % 5 == 0:   # disturb timing: competing traffic on another stream
        with torch.cuda.stream(side):
            torch.empty(1 << 26, device=dev).fill_(1.0)
    o1, l1 = run()
    bad += int(not (torch.equal(o1.view(torch.int16), o0.view(torch.int16)) and torch.equal(l1, l0)))
torch.cuda.synchronize()
print(json.dumps({"beside_gemm": gemm is not None, "iterations": N, "split_requests": int(ns[-1]) - BS, "mismatching_launches": bad,
                  "merge_counters_back_to_zero": bool((meta[:, 5:] == 0).all())}))
