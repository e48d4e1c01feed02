"""GPU box: device side of eps.fast_ep.AllToAll at world 1 (the exchange is a copy): per-kernel and whole dispatch / combine
times in a hipGraph, at the per-rank shape of BASELINE config 4 (32 tokens, top-8, 32 local experts, hidden 7168) and at a
256-token rank.  usage: python tools/time_ep.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
from eps.fast_ep import AllToAll
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)

def timed(fn, reps=20, inner=8):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(inner): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * inner)

for T in (32, 256):
    E, K, HID = 32, 8, 7168
    a2a = AllToAll(K, E, HID, T, None)
    x = torch.randn(T, HID, device=dev, generator=g).to(torch.bfloat16)
    idx = torch.stack([torch.randperm(E, device=dev, generator=g)[:K] for _ in range(T)]).to(torch.int32)
    w = torch.rand(T, K, device=dev, generator=g)
    ex = torch.empty(E + 1, dtype=torch.int32, device=dev)
    rows = torch.zeros(T * K, HID, dtype=torch.bfloat16, device=dev)
    out = torch.empty(T, HID, dtype=torch.bfloat16, device=dev)
    d = timed(lambda: a2a.dispatch(out_exclusive_sum=ex, out_expert_x=rows, dp_x=x, indices=idx, num_global_tokens=T))
    c = timed(lambda: a2a.combine(out_tokens=out, weights=w, expert_y=rows, num_global_tokens=T))
    print(json.dumps({"tokens": T, "top_k": K, "local_experts": E, "hidden": HID, "dispatch_us": round(d, 1), "combine_us": round(c, 1),
                      "kernels": "dispatch: route_dedup, send_rows, sort(+inverse), gather_rows_div; combine: gather_f32, combine (expert side), combine (home side)"}))
