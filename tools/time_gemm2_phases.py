"""GPU box: per-k-block segment cycles of the round-3 256x256 grouped-GEMM kernel (grouped_gemm_fp8_big2.hip, FL_GEMM2_TIMING
build: tools/build_gemm_exp.sh G2T "-DFL_GEMM2_TIMING -DFL_GEMM_BIG_DEFAULT=2").  usage: time_gemm2_phases.py [N] [K]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FLUENT_MI355_LIB"] = os.path.join(ROOT, "sglang-fluentllm_amd", "fluent_mi355", os.environ.get("GT_LIB", "libfluent_exp_G2T.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, numpy as np
import deep_gemm
from fluent_mi355._lib import lib
dev = torch.device("cuda:0")
E, N, K, R = int(os.environ.get("GT_E", "32")), int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 7168, int(os.environ.get("GT_ROWS", "512"))
g = torch.Generator(device=dev).manual_seed(0)
W = torch.randint(0, 120, (E, N, K), device=dev, generator=g, dtype=torch.uint8).view(torch.float8_e4m3fn)
Ws = torch.rand(E, N // 128, K // 128, device=dev, generator=g) * 1e-2
M = E * R
A = torch.randint(0, 120, (M, K), device=dev, generator=g, dtype=torch.uint8).view(torch.float8_e4m3fn)
As = torch.rand(M, K // 128, device=dev, generator=g)
ex = (torch.arange(E + 1, device=dev) * R).to(torch.int32)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
dbg = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
lib.fl_gemm2_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.fl_gemm2_debug_set_buffer(dbg.data_ptr())
for _ in range(3):
    deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg.cpu().numpy().reshape(-1, 8, 8).astype(np.float64)      # [block, wave, counter]
d = d[d[:, 0, 4] > 0]
KB = K // 128
for name, sl in (("group X (waves 0-3)", slice(0, 4)), ("group Y (waves 4-7)", slice(4, 8))):
    x = d[:, sl, :].reshape(-1, 8)
    print(f"{name}: per k block (cycles): L segments {x[:,0].mean()/KB:.0f}  M segments {x[:,1].mean()/KB:.0f}  vmcnt waits {x[:,2].mean()/KB:.0f}  "
          f"barriers {x[:,3].mean()/KB:.0f}  loop total {x[:,4].mean()/KB:.0f}")
x = d.reshape(-1, 8)
nwg = d.shape[0]
w = d[:, 0, 7]
starts = np.sort(w - w.min()) / 100.0   # us
rounds = max(1, nwg // 256)
print(f"workgroups {nwg}: epilogue issue {x[:,5].mean():.0f} cycles, k loop {x[:,4].mean():.0f}, epilogue {x[:,6].mean():.0f} (max {x[:,6].max():.0f}); "
      f"workgroup start times (us): first round ends {starts[min(255, nwg-1)]:.1f}, median {np.median(starts):.1f}, last {starts[-1]:.1f}; kernel {ms*1e3:.1f} us "
      f"-> {ms*1e3/rounds:.1f} us per round of 256")
print(f"{os.environ.get('GT_LIB','G2T')} N={N} K={K} M={M}: {ms:.3f} ms = {2.0*M*N*K/ms/1e9:.0f} TFLOP/s (timing build)")
