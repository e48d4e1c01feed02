"""GPU box: per-op device time of the EP row kernels (csrc/ep_a2a.hip) at one rank's decode shape (BASELINE config 4:
32 tokens, top-8 over 32 local experts, hidden 7168), each op replayed REP times inside one hipGraph."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
from fluent_mi355.ep import HipRowOps

dev = torch.device("cuda:0")
T, K, E, HID, REP = int(os.environ.get("T", 32)), 8, 32, 7168, 20
g = torch.Generator(device=dev).manual_seed(0)
ops = HipRowOps()
x = torch.randn(T, HID, device=dev, generator=g).to(torch.bfloat16)
idx = torch.stack([torch.randperm(E, device=dev, generator=g)[:K] for _ in range(T)]).to(torch.int32).reshape(-1)
w = torch.rand(T, K, device=dev, generator=g)
S = T * K
slot = torch.empty(S, dtype=torch.int32, device=dev); eid = torch.empty(S, dtype=torch.int32, device=dev)
buf = torch.empty(S, HID, dtype=torch.bfloat16, device=dev); rows = torch.empty_like(buf); back = torch.empty_like(buf)
order = torch.empty(S, dtype=torch.int32, device=dev); ex = torch.empty(E + 1, dtype=torch.int32, device=dev)
out = torch.empty(T, HID, dtype=torch.bfloat16, device=dev)
steps = dict(route=lambda: ops.route(idx, E, 1, S, slot, eid), send=lambda: ops.send(x, slot, K, buf),
             sort=lambda: ops.sort(eid, E, order, ex), gather=lambda: ops.gather(buf, order, S, rows),
             scatter=lambda: ops.scatter(rows, order, S, back), combine=lambda: ops.combine(back, slot, w, out, K))
res = {}
for name, fn in steps.items():
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(REP): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
    res[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / (2 * REP), 2)
print(json.dumps(dict(tokens=T, top_k=K, hidden=HID, **res)))
