"""GPU box: per-phase cycle breakdown of the MLA decode kernel (FL_MLA_TIMING build)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FLUENT_MI355_LIB"] = os.path.join(ROOT, "sglang-fluentllm_amd", "fluent_mi355", os.environ.get("TIMING_LIB", "libfluent_exp_TIMING.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench, numpy as np
import flash_mla_fp8 as fm
from fluent_mi355 import lib
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
wl = bench.build_workload(dev, 1, bench.BS, bench.SEQ, H, seed=1)
if os.environ.get("SHARE_PAGES"):   # experiment: every request reads request 0's pages (L2 hits after the first)
    k = int(os.environ["SHARE_PAGES"])
    wl["block_table"] = wl["block_table"][torch.arange(bench.BS, device=dev) // k * k].contiguous()
meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
qn, qs, qr = fm.quantize_ckv_per_token_head(wl["q"], 512)
pages = wl["pages"]
XK = H > 64
nblocks = meta.shape[0] * ((H + 127) // 128 if XK else (H + 63) // 64)
REC = 18 if H > 64 else 10   # u64 words per wave record (X kernel: 16 phases + total + last)
dbg = torch.zeros(nblocks * 4 * REC * 2, dtype=torch.int32, device=dev)
setter = lib.fl_mla_debug_set_buffer_x if XK else lib.fl_mla_debug_set_buffer
setter.argtypes = [ctypes.c_void_p]
setter(dbg.data_ptr())
k_lora, k_scale, k_rope = wl["caches"][0]
for _ in range(3):
    fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                   k_scale.view(pages, 64, 1, 1), wl["block_table"], wl["seqlens"], 512, meta, ns, bench.SCALE, True)
torch.cuda.synchronize()
d = dbg.cpu().numpy().view(np.uint64).reshape(nblocks * 4, REC).astype(np.float64)
d = d[d[:, REC - 2] > 0]   # (loader waves of the 64-row mapping leave their record empty)
if XK:
    names_x = ["barrier", "DMA issue + triples", "QK block 0", "QK block 1 || softmax 0", "PV || softmax 1", "epilogue", "vmcnt wait (page landed)", "prologue: O init",
               "prologue: sched row, lengths, window", "prologue: DMA issue p0,p1", "prologue: Q loads", "epilogue: normalisers+barrier",
               "[variant] first step (1)", "[variant] FAST steps", "[variant] generic steps", "[variant] last step (1)"]
names = ["prep(scale scratch)", "QK issue+Vt prefetch", "softmax+P publish", "waits+barrier", "DMA issue", "P fetch+O ref", "PV issue", "-"]
tiles = bench.SEQ // 64
if XK:
    names = names_x
    tiles = tiles * bench.BS / meta.shape[0]   # pages per workgroup
NP = REC - 2
tot = d[:, NP].mean()
life = d[:, NP]
if XK: print(f"lifetime min {life.min():.0f} p50 {np.percentile(life,50):.0f} p90 {np.percentile(life,90):.0f} max {life.max():.0f}; by XCD (block%8) mean: " + " ".join(f"{life.reshape(-1,4)[x::8].mean():.0f}" for x in range(8)))
print(f"H={H}: mean wave lifetime {tot:.0f} ticks; per page {tot/tiles:.0f} (s_memtime ticks, 100 MHz const clock -> x{2200/100:.0f} for ~cycles)")
for i in range(16 if XK else 7):
    print(f"  {names[i]:24s} {d[:, i].mean()/tiles:8.1f} ticks/page  ({100*d[:, i].mean()/tot:5.1f} %)")
if XK and os.environ.get("SPLIT_PARITY"):
    dd = d.reshape(-1, 4, REC)
    for i in range(16):
        print(f"    {names[i]:40s} even blocks {dd[0::2,:,i].mean()/tiles:8.1f}   odd blocks {dd[1::2,:,i].mean()/tiles:8.1f}")
print(f"  outside page loop        {(tot - d[:, :(12 if XK else NP)].sum(1).mean())/tiles:8.1f} ticks/page-equivalent")
