"""GPU box: K2-fp8 (flash_mla_with_kvcache over one plain-fp8 [.,576] cache, device-scalar descales) at the cfg2 shape:
us per launch in a hipGraph over `layers` caches.  FLUENT_MI355_LIB selects another build of the library."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_fp8 as fm
dev = torch.device("cuda:0")
BS, SEQ, H, layers = 128, 4096, 128, 8
g = torch.Generator(device=dev).manual_seed(0)
npg = SEQ // 64; pages = BS * npg + 1
# N(0,1) values cast to e4m3 (random BYTES would be values up to 240: scores of 1e6 and a second pass for every request)
caches = [torch.randn(pages, 64, 1, 576, device=dev, generator=g).to(torch.float8_e4m3fn) for _ in range(layers)]
bt = (torch.randperm(pages - 1, device=dev, generator=g).to(torch.int32) + 1).view(BS, npg).contiguous()
lens = torch.full((BS,), SEQ, dtype=torch.int32, device=dev)
q = torch.randn(BS, 1, H, 576, device=dev, generator=g).to(torch.float8_e4m3fn)
one = torch.ones(1, device=dev)
meta, ns = fm.get_mla_metadata(lens, H, 1)
def k2(l): fm.flash_mla_with_kvcache(q, caches[l], bt, lens, 512, meta, ns, bench.SCALE, True, one, one)
for l in range(layers): k2(l)
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): k2(0)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for l in range(layers): k2(l)
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): gr.replay()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 / (5 * layers)
alg = BS * (SEQ * 576 + H * 576 + H * 1024 + 4 * npg)
print(json.dumps({"kernel": "K2-fp8 plain [.,576]", "parts": int(meta.shape[0]), "us_per_launch": round(t * 1e6, 1), "GBs": round(alg / t / 1e9, 1)}))
