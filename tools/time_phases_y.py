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
RAGGED = os.environ.get("RAGGED") == "1"   # lengths uniform in seq/2 .. 3 seq/2 (bench.py's cfg2_ragged at seq = 4096)
NL = int(os.environ.get("LAYERS", "1"))   # > 1: the launches walk NL distinct caches (HBM-cold pages: what the bench step sees)
wl = bench.build_workload(dev, NL, bs, seq * 3 // 2 if RAGGED else seq, H, seed=1)
if RAGGED:
    g = torch.Generator(device=dev).manual_seed(77)
    wl["seqlens"] = torch.randint(seq // 2, seq * 3 // 2 + 1, (bs,), device=dev, generator=g, dtype=torch.int32)
if os.environ.get("SHARE_PAGES"):   # experiment: groups of k requests read the same pages (k = bs: every page an L2 hit after the first touch)
    k_ = int(os.environ["SHARE_PAGES"])
    wl["block_table"] = wl["block_table"][torch.arange(bs, device=dev) // k_ * k_].contiguous()
meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
qn, qs, qr = fm.quantize_ckv_per_token_head(wl["q"], 512)
pages = wl["pages"]
NRT = 1 if H <= 32 else 2            # row tiles per workgroup: 4-wave workgroups for <= 32 query rows per request, 8-wave otherwise
nblocks = meta.shape[0] * ((H + 32 * NRT - 1) // (32 * NRT))
REC = 14
dbg = torch.zeros(nblocks * 8 * REC * 2, dtype=torch.int32, device=dev)
lib.fl_mla_debug_set_buffer_y.argtypes = [ctypes.c_void_p]
lib.fl_mla_debug_set_buffer_y(dbg.data_ptr())
for it_ in range(3 * NL):
    k_lora, k_scale, k_rope = wl["caches"][it_ % NL]
    fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                   k_scale.view(pages, 64, 1, 1), wl["block_table"], wl["seqlens"], 512, meta, ns, bench.SCALE, True)
torch.cuda.synchronize()
d = dbg.cpu().numpy().view(np.uint64).reshape(nblocks, 8, REC).astype(np.float64)
steps = float(((wl["seqlens"] + 63) // 64).sum()) / meta.shape[0]
print(f"H={H} bs={bs} seq={seq} ragged={RAGGED}: parts {meta.shape[0]}, pieces {int(ns[-1])}, pages per part {steps:.1f}")
TICK = 10.0   # print unit: ticks x 10 (the counter runs at roughly the shader clock: read 'ns' as 0.1 ticks)
QK = ["barrier wait", "deferred sums + next scale prep (after publish)", "loop control", "LDS drain", "B_n + normalisers + E0", "request prologue",
      "  K operand reads + MFMA issue", "  half-max exchange (permlane)", "  rope/scale load issue", "  MFMA drain + scaling + max", "  exp2 + e4m3 + publish", "  (mid barrier)"]
PV = ["page-landed wait (vmcnt)", "barrier wait", "PV MFMAs + V^T reads + refill issue", "E0", "epilogue (store)", "request prologue", "  (to mid barrier)", "merge: wait for the other pieces", "merge: read + combine + store"] + ["-"] * 3
for role, sl, names in (("QK waves", slice(0, 2 * NRT), QK), ("PV waves", slice(2 * NRT, 4 * NRT), PV)):
    x = d[:, sl, :].reshape(-1, REC)
    x = x[x[:, 12] > 0]
    life = x[:, 12]
    ghz = (x[:, 12] / (x[:, 13] * 10.0)).mean()
    print(f"{role}: lifetime mean {life.mean():.0f} cycles = {x[:, 13].mean() / 100:.1f} us wall (shader clock {ghz:.2f} GHz), min {life.min():.0f} max {life.max():.0f}; {steps:.0f} steps per workgroup")
    for i in range(12):
        if names[i] != "-":
            print(f"   {names[i]:50s} {x[:, i].mean()/steps:8.1f} cycles/step  {x[:, i].mean():9.0f} per workgroup ({100*x[:, i].mean()/life.mean():5.1f} %)")
if RAGGED:
    # least-squares cost model of a workgroup's lifetime: cycles ~ a x pages + b x whole requests + c x merging pieces + d x other split pieces + e
    m = meta.cpu().numpy(); nsc = ns.cpu().numpy(); lens = wl["seqlens"].cpu().numpy(); nt = (lens + 63) // 64
    rows = []
    for p_ in range(m.shape[0]):
        br, bt_, er, et = m[p_, 0], m[p_, 1], m[p_, 2], m[p_, 3]
        pages = whole = merg = other = 0
        r, t = br, bt_
        while r < bs and (r < er or (r == er and et > 0)):
            te = nt[r] if r < er else min(et, nt[r])
            pages += te - t
            split = nsc[r + 1] - nsc[r] > 1
            if not split: whole += 1
            elif t == 0: merg += 1
            else: other += 1
            r, t = r + 1, 0
        rows.append((pages, whole, merg, other, 1.0))
    A = np.array(rows, dtype=np.float64)
    RG = (H + 63) // 64
    life = d[:, 4:8, 12].mean(axis=1)           # PV waves, per workgroup
    part_of = np.array([((b >> 3) // RG) * 8 + (b & 7) if m.shape[0] % 8 == 0 else b // RG for b in range(nblocks)])
    X = A[part_of]
    coef, *_ = np.linalg.lstsq(X, life, rcond=None)
    res = life - X @ coef
    print(f"cost model (PV wave lifetime, cycles): {coef[0]:.0f} per page + {coef[1]:.0f} per whole request + {coef[2]:.0f} per merging piece + {coef[3]:.0f} per other split piece + {coef[4]:.0f}; residual rms {res.std():.0f} (lifetime std {life.std():.0f})")
    print("parts: pages min/mean/max", A[:, 0].min(), A[:, 0].mean(), A[:, 0].max(), " pieces per part mean", A[:, 1:4].sum(axis=1).mean())
