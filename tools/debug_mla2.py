"""Debug (GPU box): dump PV operands of page 0 from the FL_MLA_DEBUG build and compare with expectations."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FLUENT_MI355_LIB"] = os.path.join(ROOT, "sglang-fluentllm_amd", "fluent_mi355", "libfluent_dbg.so")
for p in (ROOT, os.path.join(ROOT, "sglang-fluentllm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import flash_mla_fp8 as fm
from fluent_mi355 import lib
from helpers import make_paged_case
dev = torch.device("cuda:0")
SCALE = 192 ** -0.5
L, H = 64, 16
c = make_paged_case([L], H, seed=7, poison_tail=False)
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()}
pages = c["total_pages"]
qn, qs, qr = fm.quantize_ckv_per_token_head(d["q"].contiguous(), 512)
meta = torch.tensor([[0, 0, 1, 0, 0, 0, 0, 0]], dtype=torch.int32, device=dev)
ns = torch.arange(2, dtype=torch.int32, device=dev)
dbg = torch.zeros(4 * 64 * 64, dtype=torch.int32, device=dev)
lib.fl_mla_debug_set_buffer.argtypes = [ctypes.c_void_p]
print("set", lib.fl_mla_debug_set_buffer(dbg.data_ptr()))
o, lse = fm.flash_mla_ckv_fp8_per_token(qn, qr, d["k_lora"].view(pages, 64, 1, 512), d["k_rope"].view(pages, 64, 1, 64), qs,
                                        d["k_scale"].view(pages, 64, 1, 1), d["block_table"], d["cache_seqlens"], 512, meta, ns, SCALE, True)
torch.cuda.synchronize()
D = dbg.cpu().numpy().reshape(4, 64, 64)
page = int(c["block_table"][0, 0])
V = c["k_lora"].view(-1, 512)[page * 64:(page + 1) * 64].numpy()   # [64 tokens, 512] fp8 bytes
def tok(Hh, q):
    return 32 * Hh + 4 * (q >> 4) + (q & 3) + 8 * ((q & 15) >> 2)
for wave in range(4):
    rg, W = wave % 2, wave // 2
    bad_a = 0
    for lane in range(64):
        li, lh = lane & 31, lane >> 5
        a = D[wave, lane, 8:16].view(np.uint8)          # 32 bytes A operand for jb=0
        dcol = 256 * W + (li & 15) + 64 * (li >> 4)      # jb=0
        exp = np.array([V[tok(lh, q), dcol] for q in range(32)], dtype=np.uint8)
        if not np.array_equal(a, exp):
            if bad_a < 2: print(f"wave {wave} lane {lane}: A mismatch got {a[:8]} exp {exp[:8]}")
            bad_a += 1
    sb = D[wave, :, 16]; m0 = D[wave, :, 17].view(np.float32); m1 = D[wave, :, 18].view(np.float32); mo = D[wave, :, 19].view(np.float32); mw = D[wave, :, 20].view(np.float32)
    print(f"wave {wave} (rg {rg}, W {W}): A bad lanes {bad_a}; sb[:4]={sb[:4]} sb[32:36]={sb[32:36]} m0[:4]={m0[:4]} m1[:4]={m1[:4]} mo[:4]={mo[:4]} mw[:4]={mw[:4]}")
print("pb equal between wave0 and wave2:", np.array_equal(D[0, :, 0:8], D[2, :, 0:8]))
pb = D[0, :, 0:8].view(np.uint8).reshape(64, 32)
print("row0 H0 P bytes:", pb[0][:16], " row0 H1:", pb[32][:16])

# ---- emulate the PV MFMA of jb=0 from the dumped operands ----
lut = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
for wave in (0, 2):
    A = lut[D[wave, :, 8:16].view(np.uint8).reshape(64, 32)]      # [lane, q]
    B = lut[D[wave, :, 0:8].view(np.uint8).reshape(64, 32)]
    sbv = D[wave, :, 16].astype(np.float64)
    Oc = D[wave, :, 32:48].view(np.float32)                       # [lane, reg]
    exp = np.zeros((32, 32))                                      # [i (d row), n (query row)]
    for Hh in range(2):
        exp += (A[32 * Hh:32 * Hh + 32] @ (B[32 * Hh:32 * Hh + 32] * (2.0 ** (sbv[32 * Hh:32 * Hh + 32] - 127))[:, None]).T)
    got = np.zeros((32, 32))
    for lane in range(64):
        for r in range(16):
            got[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31] = Oc[lane, r]
    err = np.abs(got - exp)
    print(f"wave {wave}: PV jb0 emu: max|exp|={np.abs(exp).max():.3f} max err={err.max():.4f}; err by query row n (first 8): {[round(float(err[:, n].max()),3) for n in range(8)]}")
    # alternative hypotheses
    exp_noscale = sum(A[32*h:32*h+32] @ B[32*h:32*h+32].T for h in range(2))
    print("   vs no-scale:", round(float(np.abs(got-exp_noscale).max()),4))
    exp_swapped = sum(A[32*h:32*h+32] @ (B[32*h:32*h+32] * (2.0 ** (sbv[32*(1-h):32*(1-h)+32] - 127))[:, None]).T for h in range(2))
    print("   vs block-swapped scale:", round(float(np.abs(got-exp_swapped).max()),4))
    exp_l0 = sum(A[32*h:32*h+32] @ (B[32*h:32*h+32] * (2.0 ** (sbv[0:32] - 127))[:, None]).T for h in range(2))
    print("   vs lanes0-31 scale for both blocks:", round(float(np.abs(got-exp_l0).max()),4))

print("---- least-squares fit got[:,n] = alpha*X0 + beta*X1 per query row n (wave 0)")
wave = 0
A = lut[D[wave, :, 8:16].view(np.uint8).reshape(64, 32)]
B = lut[D[wave, :, 0:8].view(np.uint8).reshape(64, 32)]
Oc = D[wave, :, 32:48].view(np.float32)
got = np.zeros((32, 32))
for lane in range(64):
    for r in range(16):
        got[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31] = Oc[lane, r]
sbv = D[wave, :, 16]
for n in range(16):
    X0 = A[0:32] @ B[n]; X1 = A[32:64] @ B[32 + n]
    M = np.stack([X0, X1], 1)
    coef, res, *_ = np.linalg.lstsq(M, got[:, n], rcond=None)
    print(f"n={n:2d} sb=({sbv[n]},{sbv[32+n]}) alpha={coef[0]:.4f} beta={coef[1]:.4f} resid={float(np.abs(M@coef-got[:,n]).max()):.2f}")

print("---- hypothesis: MX block b = bytes [16b,16b+16) of BOTH lane halves; scale from lane n+32b")
s = 2.0 ** (sbv.astype(np.float64) - 127)
exp2 = np.zeros((32, 32))
for n in range(32):
    lo = A[0:32, :16] @ B[n, :16] + A[32:64, :16] @ B[32 + n, :16]
    hi = A[0:32, 16:] @ B[n, 16:] + A[32:64, 16:] @ B[32 + n, 16:]
    exp2[:, n] = s[n] * lo + s[32 + n] * hi
print("max err:", float(np.abs(got - exp2).max()), " max|exp|:", float(np.abs(exp2).max()))
