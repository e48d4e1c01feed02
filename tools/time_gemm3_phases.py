"""GPU box: per-k-block phase cycles of the round-6 192 x 256 grouped-GEMM kernel (grouped_gemm_fp8_big3.hip, FL_GEMM3_TIMING build:
tools/build_gemm3_var.sh G3T -DFL_GEMM3_TIMING).  usage: time_gemm3_phases.py [N] [K]   env: GT_E experts, GT_ROWS rows per expert, GT_LIB"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FLUENT_MI355_LIB"] = os.path.join(ROOT, "sglang-fluentllm_amd", "fluent_mi355", os.environ.get("GT_LIB", "libfluent_exp_G3T.so"))
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
dbg = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device=dev)
lib.fl_gemm3_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.fl_gemm3_debug_set_buffer(dbg.data_ptr())
for _ in range(3):
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)      # [tile slot, wave, counter]
d = d[d[:, 0, 4] > 0]
KB = K // 128
x = d.reshape(-1, 8)
print(f"tiles {d.shape[0]}: per k block (cycles, mean over waves): MFMA slots + fillers {x[:,0].mean()/KB:.0f}  vmcnt waits {x[:,1].mean()/KB:.0f}  "
      f"barriers {x[:,2].mean()/KB:.0f}  lgkm wait {x[:,3].mean()/KB:.0f}  k loop total {x[:,4].mean()/KB:.0f} (ideal 1536 = 24 MFMAs x 64);  "
      f"epilogue {x[:,5].mean():.0f} cycles per tile, k loop {x[:,4].mean():.0f}")
for w in range(4):
    y = d[:, w, :]
    print(f"  wave {w}: slots {y[:,0].mean()/KB:.0f} vmcnt {y[:,1].mean()/KB:.0f} barrier {y[:,2].mean()/KB:.0f} lgkm {y[:,3].mean()/KB:.0f}")
tiles = d.shape[0]
print(f"{os.environ.get('GT_LIB','G3T')} N={N} K={K} M={M} ({E} x {R}): {ms:.3f} ms = {2.0*M*N*K/ms/1e9:.0f} TFLOP/s (timing build); "
      f"clock from cycles: {(x[:,4].mean()+x[:,5].mean())*np.ceil(tiles/256)/ (ms*1e3):.0f} MHz (rough)")
