"""GPU box: K7 all-layer KV row move (fluent_mi355.kvmove) vs the reference's native per-buffer indexing loop
(memory_pool.py:756-763 restated with torch ops on the device), 61 layers x (512 B, 4 B, 128 B rows), hipGraph replays."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
from fluent_mi355.kvmove import KVMoveTable

dev = torch.device("cuda:0")
S, L = 1 << 16, 61
bufs = []
for _ in range(L):
    bufs += [torch.randint(0, 255, (S, 1, 512), dtype=torch.uint8, device=dev), torch.rand(S, 1, 1, device=dev),
             torch.randn(S, 1, 64, device=dev).to(torch.bfloat16)]
table = KVMoveTable(bufs)
out = {}
for n in (64, 256, 4096):
    src = (torch.randperm(S // 2, device=dev)[:n] + S // 2).to(torch.int64)
    tgt = torch.randperm(S // 2, device=dev)[:n].to(torch.int64)
    def hip(): table.move(tgt, src)
    def native():
        for b in bufs: b[tgt] = b[src]
    for name, fn in (("hip", hip), ("torch_indexing", native)):
        fn(); torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): fn()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); torch.cuda.synchronize()
        out[f"{name}_n{n}_us"] = round(e0.elapsed_time(e1) * 200, 1)
print(json.dumps(out))
