"""GPU box: per-phase breakdown of the K2-bf16 kernel (mla_decode_bf16.hip built with -DFL_MLA_TIMING: tools/build_exp_bf16.sh).
usage: python tools/time_phases_bf16.py [H] [bs] [seq]   (SMALLSET=n: every request reads the same n pages)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FLUENT_MI355_LIB"] = os.path.join(ROOT, "sglang-fluentllm_amd", "fluent_mi355", "libfluent_exp_bf16_TIMING.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench, numpy as np
import flash_mla_swap as fsw
from fluent_mi355 import lib
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 128
SEQ = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g = torch.Generator(device=dev).manual_seed(0)
npg = SEQ // 64; pages = BS * npg + 1
caches = [torch.randn(pages, 64, 1, 576, device=dev, generator=g).to(torch.bfloat16) for _ in range(2)]
bt = (torch.randperm(pages - 1, device=dev, generator=g).to(torch.int32) + 1).view(BS, npg).contiguous()
if os.environ.get("SMALLSET"): bt = (bt % int(os.environ["SMALLSET"]) + 1).contiguous()
lens = torch.full((BS,), SEQ, dtype=torch.int32, device=dev)
q = torch.randn(BS, 1, H, 576, device=dev, generator=g).to(torch.bfloat16)
meta, ns = fsw.get_mla_metadata(lens, H, 1)
NRT = 1 if H <= 32 else 2
nblocks = meta.shape[0] * ((H + 32 * NRT - 1) // (32 * NRT))
REC = 10
dbg = torch.zeros(nblocks * 8 * REC * 2, dtype=torch.int32, device=dev)
lib.fl_mla_debug_set_buffer_b.argtypes = [ctypes.c_void_p]
lib.fl_mla_debug_set_buffer_b(dbg.data_ptr())
for it in range(4): fsw.flash_mla_with_kvcache(q, caches[it % 2], bt, lens, 512, meta, ns, bench.SCALE, True)
torch.cuda.synchronize()
d = dbg.cpu().numpy().view(np.uint64).reshape(nblocks, 8, REC).astype(np.float64)
steps = SEQ / 32 * BS / meta.shape[0]
print(f"H={H} bs={BS} seq={SEQ}: parts {meta.shape[0]}, workgroups {nblocks} x {4 * NRT} waves, {steps:.0f} tile steps per workgroup")
QK = ["(loop control)", "barrier wait", "K reads + 36 MFMAs", "-", "softmax + P store", "request prologue (Q load, R0)", "B_n + normalisers + E0", "-"]
PV = ["landed wait (vmcnt)", "barrier wait", "tail zero + P / f read", "16 MFMAs + V^T reads", "refill issue + block-table read", "request prologue (R0)", "E0", "epilogue (store)"]
for role, sl, names in (("QK waves", slice(0, 2 * NRT), QK), ("PV waves", slice(2 * NRT, 4 * NRT), PV)):
    x = d[:, sl, :].reshape(-1, REC)
    x = x[x[:, 8] > 0]
    ghz = (x[:, 8] / (x[:, 9] * 10.0)).mean()
    print(f"{role}: lifetime mean {x[:, 8].mean():.0f} cycles = {x[:, 9].mean() / 100:.1f} us wall (shader clock {ghz:.2f} GHz)")
    for i in range(8):
        if names[i] != "-": print(f"   {names[i]:42s} {x[:, i].mean() / steps:8.0f} cycles per step  ({100 * x[:, i].mean() / x[:, 8].mean():5.1f} %)")
