"""Profiling target (GPU box): the bench workload's K1 kernel only (few layers), for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import flash_mla_fp8 as fm

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H = int(sys.argv[3]) if len(sys.argv) > 3 else bench.H
bs = int(sys.argv[4]) if len(sys.argv) > 4 else bench.BS
seq = int(sys.argv[5]) if len(sys.argv) > 5 else bench.SEQ
dev = torch.device("cuda:0")
wl = bench.build_workload(dev, layers, bs, seq, H, seed=1)
meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
qn, qs, qr = fm.quantize_ckv_per_token_head(wl["q"], 512)
pages = wl["pages"]
for _ in range(reps):
    for l in range(layers):
        k_lora, k_scale, k_rope = wl["caches"][l]
        fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                       k_scale.view(pages, 64, 1, 1), wl["block_table"], wl["seqlens"], 512, meta, ns, bench.SCALE, True)
torch.cuda.synchronize()
print("done")
