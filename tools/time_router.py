"""GPU box: device time of R1 (moe_gate_kernel) per call at decode / prefill token counts, inside one hipGraph."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, flashinfer
dev = torch.device("cuda:0"); out = {}
bias = torch.randn(256, device=dev) * 0.1
for T in (32, 256, 4096, 16384):
    x = torch.randn(T, 256, device=dev)
    f = lambda: flashinfer.moe_fused_gate(x, bias, 8, 4, 8, 0, 2.5, True)
    f(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): f()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    out[f"T{T}_us"] = round(e0.elapsed_time(e1) * 50, 2)
print(json.dumps(out))
