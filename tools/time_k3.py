"""GPU box: device time of K3 (mla_metadata_kernel) per call, replayed inside one hipGraph."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import flash_mla_fp8 as fm

dev = torch.device("cuda:0")
out = {}
for bs, seq, rows in ((1, 4096, 128), (8, 4096, 128), (128, 4096, 128), (256, 8192, 16), (32, 8192, 128), (2048, 1024, 16)):
    g = torch.Generator().manual_seed(bs)
    sl = (torch.randint(seq // 2, seq + 1, (bs,), generator=g, dtype=torch.int32)).to(dev)
    fm.get_mla_metadata(sl, rows, 1); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fm.get_mla_metadata(sl, rows, 1)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): fm.get_mla_metadata(sl, rows, 1)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    out[f"bs{bs}_seq{seq}_rows{rows}_us"] = round(e0.elapsed_time(e1) * 100, 1)
print(json.dumps(out))
