"""GPU box: per-phase breakdown of the <= 32-row MLA decode kernel (mla_decode_fp8.hip, FL_MLA_TIMING build: tools/build_exp.sh TIMING).
usage: python tools/time_phases_small.py [H] [bs] [seq]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FLUENT_MI355_LIB"] = os.path.join(ROOT, "sglang-fluentllm_amd", "fluent_mi355", os.environ.get("TIMING_LIB", "libfluent_exp_TIMING.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench, numpy as np
import flash_mla_fp8 as fm
from fluent_mi355 import lib
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bs = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BS
seq = int(sys.argv[3]) if len(sys.argv) > 3 else bench.SEQ
NL = int(os.environ.get("LAYERS", "16"))
wl = bench.build_workload(dev, NL, bs, seq, H, seed=1)
meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
qn, qs, qr = fm.quantize_ckv_per_token_head(wl["q"], 512)
pages = wl["pages"]
nblocks = meta.shape[0] * ((H + 31) // 32)
REC = 10
dbg = torch.zeros(nblocks * 4 * REC * 2, dtype=torch.int32, device=dev)
lib.fl_mla_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.fl_mla_debug_set_buffer(dbg.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it_ in range(3 * NL):
    if it_ == NL: e0.record()
    k_lora, k_scale, k_rope = wl["caches"][it_ % NL]
    fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                   k_scale.view(pages, 64, 1, 1), wl["block_table"], wl["seqlens"], 512, meta, ns, bench.SCALE, True)
e1.record(); torch.cuda.synchronize()
d = dbg.cpu().numpy().view(np.uint64).reshape(nblocks, 4, REC).astype(np.float64)
steps = float(((wl["seqlens"] + 63) // 64).sum()) / meta.shape[0]
print(f"H={H} bs={bs} seq={seq}: parts {meta.shape[0]}, pieces {int(ns[-1])}, pages per part {steps:.1f}; eager launches {e0.elapsed_time(e1) * 1e3 / (2 * NL):.1f} us each (decode + combine)")
names = ["prep (tail fill)", "QK operand reads + MFMA issue + V^T prefetch", "MFMA drain + softmax + P publish", "page-landed wait + LDS drain + barrier",
         "P / reference read", "O reference update", "PV MFMAs + V^T reads (+ next page's scale prep, QK prefetch)", "-"]
x = d[:, 0:2, :].reshape(-1, REC)       # compute waves
x = x[x[:, 8] > 0]
life = x[:, 8]
print(f"compute waves: lifetime mean {life.mean():.0f} cycles (min {life.min():.0f} max {life.max():.0f}); {steps:.0f} steps per workgroup -> {life.mean() / steps:.0f} cycles per step all in")
tot = 0.0
for i in range(7):
    tot += x[:, i].mean()
    print(f"   {names[i]:72s} {x[:, i].mean() / steps:8.1f} cycles/step ({100 * x[:, i].mean() / life.mean():5.1f} %)")
print(f"   {'outside the page loop (request prologue + epilogue)':72s} {(life.mean() - tot):8.0f} cycles per workgroup ({100 * (life.mean() - tot) / life.mean():5.1f} %)")
