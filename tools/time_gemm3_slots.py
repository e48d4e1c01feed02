"""GPU box: cycle stamps in front of every MFMA of one k block per tile (grouped_gemm_fp8_big3.hip, FL_GEMM3_SLOTS build:
tools/build_gemm3_var.sh G3S -DFL_GEMM3_SLOTS [+ bounding switches]).  Prints the mean cycles per MFMA slot of the two half steps.
usage: time_gemm3_slots.py [N] [K]   env GT_LIB (default libfluent_exp_G3S.so), GT_E, GT_ROWS"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FLUENT_MI355_LIB"] = os.path.join(ROOT, "sglang-fluentllm_amd", "fluent_mi355", os.environ.get("GT_LIB", "libfluent_exp_G3S.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, numpy as np
import deep_gemm
from fluent_mi355._lib import lib
dev = torch.device("cuda:0")
E, N, K, R = int(os.environ.get("GT_E", "32")), int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 7168, int(os.environ.get("GT_ROWS", "576"))
g = torch.Generator(device=dev).manual_seed(0)
W = torch.randint(0, 120, (E, N, K), device=dev, generator=g, dtype=torch.uint8).view(torch.float8_e4m3fn)
Ws = torch.rand(E, N // 128, K // 128, device=dev, generator=g) * 1e-2
M = E * R
A = torch.randint(0, 120, (M, K), device=dev, generator=g, dtype=torch.uint8).view(torch.float8_e4m3fn)
As = torch.rand(M, K // 128, device=dev, generator=g)
ex = (torch.arange(E + 1, device=dev) * R).to(torch.int32)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
dbg = torch.zeros(8192 * 4 * 8 + 2048 * 4 * 26, dtype=torch.int64, device=dev)
lib.fl_gemm3_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.fl_gemm3_debug_set_buffer(dbg.data_ptr())
for _ in range(3):
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg[8192 * 4 * 8:].cpu().numpy().reshape(-1, 4, 26).astype(np.float64)
d = d[d[:, 0, 0] > 0]
x = d.reshape(-1, 26)
for name, o in (("even half step", 0), ("odd half step", 13)):
    dt = x[:, o + 1:o + 13] - x[:, o:o + 12]
    print(f"{name}: cycles from MFMA s to MFMA s+1 (s = 0..11; the last: to the end of the step): " + " ".join(f"{v:.0f}" for v in dt.mean(0)) + f"   sum {dt.mean(0).sum():.0f}")
gap = x[:, 13] - x[:, 12]
print(f"between the half steps {gap.mean():.0f}; k block (stamp 0 -> end of odd step) {(x[:,25]-x[:,0]).mean():.0f} cycles; tiles sampled {d.shape[0]}")
print(f"{os.environ.get('GT_LIB','G3S')} N={N} K={K} M={M}: {ms:.3f} ms = {2.0*M*N*K/ms/1e9:.0f} TFLOP/s (stamp build)")
