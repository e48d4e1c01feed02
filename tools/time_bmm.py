"""GPU box: the weight-absorption bmm's (B1 / B2, csrc/bmm_bf16.hip) at decode sizes, hipGraph of 8 calls.
usage: python tools/time_bmm.py [T ...]   (FLUENT_MI355_LIB selects another build of the library)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
from fluent_mi355.bmm import bmm
dev = torch.device("cuda:0")
H = 128
g = torch.Generator(device=dev).manual_seed(0)
w = (torch.randn(H, 256, 512, device=dev, generator=g) * 0.05).to(torch.bfloat16)
wkc = w[:, :128].transpose(1, 2).contiguous().transpose(1, 2)
wvc = w[:, 128:].contiguous().transpose(1, 2)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(8): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 160
import flashinfer
wr = (torch.randn(256, 7168, device=dev, generator=g) * 0.02).to(torch.bfloat16)
for T in (int(a) for a in (sys.argv[1:] or ["16", "128", "256"])):
    xr = torch.randn(T, 7168, device=dev, generator=g).to(torch.bfloat16)
    tr = timeit(lambda: flashinfer.dsv3_router_gemm(xr, wr, out_dtype=torch.float32))
    tl = timeit(lambda: torch.nn.functional.linear(xr, wr))
    q = torch.randn(T, H, 192, device=dev, generator=g).to(torch.bfloat16)
    Q = torch.empty(T, H, 576, dtype=torch.bfloat16, device=dev)
    att = torch.randn(T, H, 512, device=dev, generator=g).to(torch.bfloat16)
    tq = timeit(lambda: bmm(q[..., :128].transpose(0, 1), wkc, out=Q[..., :512].transpose(0, 1)))
    tv = timeit(lambda: bmm(att.transpose(0, 1), wvc))
    tt = timeit(lambda: torch.bmm(att.transpose(0, 1), wvc))
    print(json.dumps({"T": T, "H": H, "q_absorb_us": round(tq, 1), "v_absorb_us": round(tv, 1),
                      "torch_bmm_v_absorb_us": round(tt, 1),
                      "router_gemm_f32_us": round(tr, 1), "torch_linear_bf16_router_us": round(tl, 1)}))
