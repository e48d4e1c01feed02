import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import deep_gemm
import test_gemm_gpu as T
from oracle import gemm_ref
DEV = T.DEV
counts = [int(x) for x in sys.argv[1].split(",")]
N, K = int(sys.argv[2]), int(sys.argv[3])
xq, xs, W, Ws, ex = T.make_group_case(counts, N, K, seed=5)
M = xq.shape[0]
out = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device=DEV)
print("launch", counts, N, K, flush=True)
deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((xq.to(DEV), xs.to(DEV)), (W.to(DEV), Ws.to(DEV)), out[:M], ex.to(DEV), use_pdl=True)
torch.cuda.synchronize()
print("done", flush=True)
ref = gemm_ref.grouped_gemm_offset(xq, xs, W, Ws, ex)
o = out[:M].cpu().float()
print("finite", bool(torch.isfinite(o).all()), "rel", T.rel_mae(out[:M].cpu(), ref), flush=True)
err = (o - ref.float()).abs()
# where are the errors: per 32-row x 32-col block
for a in range(0, M, 32):
    row = []
    for b in range(0, N, 32):
        blk = err[a:a + 32, b:b + 32]
        row.append("." if float(blk.max()) < 0.05 * float(ref.float().abs().mean()) + 1e-2 else "X")
    print(f"{a:5d} " + "".join(row))
r = ref.float()
for (a, b) in ((0, 0), (32, 0), (64, 128), (128, 0)):
    if a < M:
        q = (o[a:a + 4, b:b + 6] / r[a:a + 4, b:b + 6])
        print("block", a, b, "out/ref:\n", q)
        print(" out", o[a, b:b + 6], "\n ref", r[a, b:b + 6])
# does a wrong row equal some other reference row?
a = 0
d = (r[:192, :64] - o[a, :64]).abs().sum(1)
print("row", a, "closest ref row", int(d.argmin()), float(d.min()))
