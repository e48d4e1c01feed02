"""GPU box: timing of the bf16 K2 kernel (flash_mla_with_kvcache, bf16 cache) at bs=128, seq=4096 (graph replay)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, flash_mla_swap as fm
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs, seq = 128, 4096
dev = torch.device("cuda:0")
pages = bs * seq // 64 + 1
layers = 4
g = torch.Generator(device=dev).manual_seed(0)
caches = [(torch.randn(pages, 64, 1, 576, device=dev, generator=g) * 0.5).to(torch.bfloat16) for _ in range(layers)]
q = torch.randn(bs, 1, H, 576, device=dev, generator=g).to(torch.bfloat16)
perm = (torch.randperm(pages - 1, device=dev, generator=g) + 1).to(torch.int32).view(bs, seq // 64)
seqlens = torch.full((bs,), seq, dtype=torch.int32, device=dev)
meta, ns = fm.get_mla_metadata(seqlens, H, 1)
def run(l): return fm.flash_mla_with_kvcache(q, caches[l], perm, seqlens, 512, meta, ns, 576 ** -0.5, True)
for l in range(layers): run(l)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for l in range(layers): run(l)
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): gr.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (10 * layers)
alg = bs * seq * 1152 + bs * H * (1152 + 1024)
print(f"bf16 K2 H={H} bs={bs} seq={seq}: {us:.1f} us/launch  {alg/us/1e3:.0f} GB/s ({alg/us/1e3/8000*100:.1f}% of 8 TB/s)")
