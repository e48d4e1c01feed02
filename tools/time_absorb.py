"""GPU box: the query side of the decode layer before K1 — four launches (B2 bmm, R2 rope, K5 + K4) vs the fused launch (A2,
csrc/mla_absorb.hip), hipGraph of 8 calls.  usage: python tools/time_absorb.py [T ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import flash_mla_fp8 as fm
from fluent_mi355.bmm import bmm
from fluent_mi355.rope import apply_rope_with_cos_sin_cache_inplace
dev = torch.device("cuda:0")
H = int(os.environ.get("H", "128"))
g = torch.Generator(device=dev).manual_seed(0)
w_kc = (torch.randn(H, 512, 128, device=dev, generator=g) * 0.05).to(torch.bfloat16).transpose(1, 2)
cache = torch.randn(8192, 64, device=dev, generator=g)
slots = 128 * 64 * 8
kc = [torch.zeros(slots, 1, 512, dtype=torch.uint8, device=dev).view(torch.float8_e4m3fn), torch.zeros(slots, 1, 1, device=dev),
      torch.zeros(slots, 1, 64, dtype=torch.bfloat16, device=dev)]
def timeit(fn):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(8): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 160
for T in (int(a) for a in (sys.argv[1:] or ["1", "16", "128", "256"])):
    q = torch.randn(T, H, 192, device=dev, generator=g).to(torch.bfloat16)
    lat = torch.randn(T, 576, device=dev, generator=g).to(torch.bfloat16)
    pos = torch.randint(0, 8192, (T,), device=dev, generator=g)
    loc = (torch.arange(T, device=dev, dtype=torch.int32) * 61 + 5)
    Q = torch.empty(T, H, 576, dtype=torch.bfloat16, device=dev)
    K = lat.unsqueeze(1)
    def chain():
        bmm(q[..., :128].transpose(0, 1), w_kc, out=Q[..., :512].transpose(0, 1))
        apply_rope_with_cos_sin_cache_inplace(pos, q[..., 128:], K[..., 512:], 64, cache, is_neox=False, output_q_rope=Q[..., 512:])
        return fm.quantize_q_and_cache_k(Q, K, kc[0], kc[1], kc[2], loc, 512)
    def fused():
        return fm.absorb_rope_quant(q, w_kc, pos, cache, lat, kc[0], kc[1], kc[2], loc)
    print(json.dumps({"T": T, "H": H, "four_launches_us": round(timeit(chain), 1), "fused_us": round(timeit(fused), 1)}))
