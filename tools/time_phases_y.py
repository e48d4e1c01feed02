"""GPU box: per-phase breakdown of the role-specialised MLA decode kernel (mla_decode_fp8_y.hip, FL_MLA_TIMING build:
tools/build_exp.sh TIMING).  usage: python tools/time_phases_y.py [H] [bs] [seq]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FLUENT_MI355_LIB"] = os.path.join(ROOT, "sglang-fluentllm_amd", "fluent_mi355", os.environ.get("TIMING_LIB", "libfluent_exp_TIMING.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench, numpy as np
import flash_mla_fp8 as fm
from fluent_mi355 import lib
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BS
seq = int(sys.argv[3]) if len(sys.argv) > 3 else bench.SEQ
wl = bench.build_workload(dev, 1, bs, seq, H, seed=1)
meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
qn, qs, qr = fm.quantize_ckv_per_token_head(wl["q"], 512)
pages = wl["pages"]
nblocks = meta.shape[0] * ((H + 63) // 64)
REC = 14
dbg = torch.zeros(nblocks * 8 * REC * 2, dtype=torch.int32, device=dev)
lib.fl_mla_debug_set_buffer_y.argtypes = [ctypes.c_void_p]
lib.fl_mla_debug_set_buffer_y(dbg.data_ptr())
k_lora, k_scale, k_rope = wl["caches"][0]
for _ in range(3):
    fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                   k_scale.view(pages, 64, 1, 1), wl["block_table"], wl["seqlens"], 512, meta, ns, bench.SCALE, True)
torch.cuda.synchronize()
d = dbg.cpu().numpy().view(np.uint64).reshape(nblocks, 8, REC).astype(np.float64)
steps = (seq // 64) * bs / meta.shape[0]
TICK = 10.0   # print unit: ticks x 10 (the counter runs at roughly the shader clock: read 'ns' as 0.1 ticks)
QK = ["barrier wait", "deferred sums + next scale prep (after publish)", "loop control", "LDS drain", "B_n + normalisers + E0", "request prologue",
      "  K operand reads + MFMA issue", "  half-max exchange (permlane)", "  rope/scale load issue", "  MFMA drain + scaling + max", "  exp2 + e4m3 + publish", "  (mid barrier)"]
PV = ["page-landed wait (vmcnt)", "barrier wait", "PV MFMAs + V^T reads + refill issue", "E0", "epilogue (store)", "request prologue", "  (to mid barrier)"] + ["-"] * 5
for role, sl, names in (("QK waves", slice(0, 4), QK), ("PV waves", slice(4, 8), PV)):
    x = d[:, sl, :].reshape(-1, REC)
    x = x[x[:, 12] > 0]
    life = x[:, 12]
    ghz = (x[:, 12] / (x[:, 13] * 10.0)).mean()
    print(f"{role}: lifetime mean {life.mean():.0f} cycles = {x[:, 13].mean() / 100:.1f} us wall (shader clock {ghz:.2f} GHz), min {life.min():.0f} max {life.max():.0f}; {steps:.0f} steps per workgroup")
    for i in range(12):
        if names[i] != "-":
            print(f"   {names[i]:50s} {x[:, i].mean()/steps:8.1f} cycles/step  ({100*x[:, i].mean()/life.mean():5.1f} %)")
