"""GPU box: K2-bf16 (flash_mla_swap.flash_mla_with_kvcache over one bf16 [.,576] cache) — us per launch in a hipGraph of `launches` launches over
`caches` distinct caches.  usage: python tools/time_k2_bf16.py [H] [bs] [seq]   (FLUENT_MI355_LIB selects another build of the library)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_swap as fsw
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 128
SEQ = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
caches_n, launches = int(os.environ.get("CACHES", "12")), int(os.environ.get("LAUNCHES", "48"))
g = torch.Generator(device=dev).manual_seed(0)
npg = SEQ // 64; pages = BS * npg + 1
base = torch.randn(pages, 64, 1, 576, device=dev, generator=g)
caches = [torch.roll(base, 7 * l, 0).to(torch.bfloat16) for l in range(caches_n)]
del base
bt = (torch.randperm(pages - 1, device=dev, generator=g).to(torch.int32) + 1).view(BS, npg).contiguous()
if os.environ.get("SMALLSET"):   # every request reads the same few pages (L2-resident): separates memory latency from the rest of the step
    bt = (bt % int(os.environ["SMALLSET"]) + 1).contiguous()
lens = torch.full((BS,), SEQ, dtype=torch.int32, device=dev)
q = torch.randn(BS, 1, H, 576, device=dev, generator=g).to(torch.bfloat16)
meta, ns = fsw.get_mla_metadata(lens, H, 1)
def k2(l): fsw.flash_mla_with_kvcache(q, caches[l], bt, lens, 512, meta, ns, bench.SCALE, True)
for l in range(caches_n): k2(l)
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): k2(0)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for i in range(launches): k2(i % caches_n)
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): gr.replay()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 / (5 * launches)
alg = BS * (SEQ * 1152 + H * 1152 + H * 1024 + 4 * npg)
print(json.dumps({"kernel": "K2-bf16 [.,576]", "H": H, "bs": BS, "seq": SEQ, "parts": int(meta.shape[0]), "us_per_launch": round(t * 1e6, 1),
                  "GBs": round(alg / t / 1e9, 1), "hbm_frac": round(alg / t / 1e9 / 8000, 4), "tag": os.environ.get("FLUENT_MLA_LIB_TAG", "")}))
