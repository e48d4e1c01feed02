"""GPU box, context measurement only (nothing in the product path calls a library GEMM): what the vendor library (hipBLASLt through
torch._scaled_mm) sustains on this box for ONE dense fp8 GEMM with the flops of the grouped w13 / w2 problems of bench.py (tensor-wise
scales: no 1x128 / 128x128 block rescale, no expert boundaries) — the practical ceiling for random operands under the power cap.
usage: python tools/ref_hipblaslt_fp8.py"""
import json, torch
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def run(M, N, K, zeros=False):
    if zeros:
        a = torch.zeros(M, K, device=dev).to(torch.float8_e4m3fn); b = torch.zeros(N, K, device=dev).to(torch.float8_e4m3fn)
    else:
        a = (torch.randn(M, K, device=dev, generator=g)).to(torch.float8_e4m3fn)
        b = (torch.randn(N, K, device=dev, generator=g)).to(torch.float8_e4m3fn)
    sa = torch.tensor(1.0, device=dev); sb = torch.tensor(1.0, device=dev)
    f = lambda: torch._scaled_mm(a, b.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return {"M": M, "N": N, "K": K, "operands": "zeros" if zeros else "N(0,1) -> e4m3", "ms": round(ms, 3), "TFLOPs": round(2.0 * M * N * K / ms / 1e9, 1)}
for args in ((131072, 4096, 7168), (131072, 7168, 2048), (16384, 4096, 7168), (8192, 8192, 8192)):
    try:
        print(json.dumps(run(*args)))
    except Exception as ex:
        print(json.dumps({"shape": args, "error": f"{type(ex).__name__}: {ex}"[:300]}))
try:
    print(json.dumps(run(131072, 4096, 7168, zeros=True)))
except Exception as ex:
    print(json.dumps({"error": str(ex)[:200]}))
# the grouped problem itself, as the library would run it: one GEMM per expert ([rows x 7168] x [7168 x 4096], weights read ONCE from HBM,
# reused by rows/256 row tiles only), E of them captured in one hipGraph
def grouped(E, rows, N, K):
    a = torch.randn(E * rows, K, device=dev, generator=g).to(torch.float8_e4m3fn)
    w = torch.empty(E, N, K, dtype=torch.float8_e4m3fn, device=dev)
    for e in range(0, E, 16):
        w[e:e + 16] = torch.randn(min(16, E - e), N, K, device=dev, generator=g).to(torch.float8_e4m3fn)
    sa = torch.tensor(1.0, device=dev); sb = torch.tensor(1.0, device=dev)
    out = torch.empty(E * rows, N, dtype=torch.bfloat16, device=dev)
    def f():
        for e in range(E):
            torch._scaled_mm(a[e * rows:(e + 1) * rows], w[e].t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16, out=out[e * rows:(e + 1) * rows])
    f(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): f()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gr.replay()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    return {"grouped_as": f"{E} library GEMMs [{rows} x {K}] x [{K} x {N}] in one hipGraph", "ms": round(ms, 3), "TFLOPs": round(2.0 * E * rows * N * K / ms / 1e9, 1)}
for args in ((256, 512, 4096, 7168), (256, 512, 7168, 2048), (256, 1024, 4096, 7168)):
    try:
        print(json.dumps(grouped(*args)))
    except Exception as ex:
        print(json.dumps({"shape": args, "error": f"{type(ex).__name__}: {ex}"[:300]}))
