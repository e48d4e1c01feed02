"""Timing helper (GPU box): K1 only on the bench workload; FLUENT_MI355_LIB selects an experimental build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_fp8 as fm
layers = int(os.environ.get("LAYERS", "8"))
H = int(sys.argv[1]) if len(sys.argv) > 1 else bench.H
bs = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BS
seq = int(sys.argv[3]) if len(sys.argv) > 3 else bench.SEQ
dev = torch.device("cuda:0")
wl = bench.build_workload(dev, layers, bs, seq, H, seed=1)
if os.environ.get("SHARE_PAGES"):   # experiment: every request reads request 0's pages (100 % L2 hits after the first)
    k = int(os.environ["SHARE_PAGES"])
    wl["block_table"] = wl["block_table"][torch.arange(bs, device=dev) // k * k].contiguous()
if os.environ.get("SEQ_PAGES"):    # experiment: pages in address order instead of randomly permuted (TLB / DRAM locality)
    wl["block_table"] = (torch.arange(bs * (seq // 64), device=dev, dtype=torch.int32) + 1).view(bs, -1).contiguous()
if os.environ.get("QSCALE"):   # experiment: logits QSCALE x wider (a later block's reference far above the first block's: the O-reference lift / redo pass)
    wl["q"] = (wl["q"].float() * float(os.environ["QSCALE"])).to(torch.bfloat16)
meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
qn, qs, qr = fm.quantize_ckv_per_token_head(wl["q"], 512)
pages = wl["pages"]
def k1(l):
    k_lora, k_scale, k_rope = wl["caches"][l]
    fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                   k_scale.view(pages, 64, 1, 1), wl["block_table"], wl["seqlens"], 512, meta, ns, bench.SCALE, True)
for l in range(layers): k1(l)
torch.cuda.synchronize()
# one hipGraph of all layers: eager launches of this op are CPU-bound (~170 us per call) and hide kernel differences
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for l in range(layers): k1(l)
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(g):
    for l in range(layers): k1(l)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps): g.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (reps * layers)
alg = bench.algorithmic_bytes(bs, seq, H, 1)
tag = "default"
print(f"{os.environ.get('FLUENT_MLA_LIB_TAG', tag)} H={H} bs={bs} seq={seq}: {us:.1f} us/launch (graph replay)  {alg/us/1e3:.0f} GB/s ({alg/us/1e3/8000*100:.1f}% of 8 TB/s)")
