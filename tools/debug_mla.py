"""Debug helper (GPU box): run K1 on tiny cases with a single scheduler part and print the error structure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sglang-fluentllm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import flash_mla_fp8 as fm
from oracle import mla_ref
from helpers import make_paged_case

dev = torch.device("cuda:0")
SCALE = 192 ** -0.5
for lens, H in (([32], 16), ([64], 16), ([128], 16), ([192], 16), ([256], 64)):
    c = make_paged_case(lens, H, seed=7, poison_tail=False)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()}
    pages = c["total_pages"]
    qn, qs, qr = fm.quantize_ckv_per_token_head(d["q"].contiguous(), 512)
    # single part: [begin_req=0, begin_tile=0, end_req=bs, end_tile=0, split=0]
    bs = len(lens)
    meta = torch.tensor([[0, 0, bs, 0, 0, 0, 0, 0]], dtype=torch.int32, device=dev)
    ns = torch.arange(bs + 1, dtype=torch.int32, device=dev)
    o, lse = fm.flash_mla_ckv_fp8_per_token(qn, qr, d["k_lora"].view(pages, 64, 1, 512), d["k_rope"].view(pages, 64, 1, 64), qs,
                                            d["k_scale"].view(pages, 64, 1, 1), d["block_table"], d["cache_seqlens"], 512, meta, ns, SCALE, True)
    torch.cuda.synchronize()
    ref, rlse = mla_ref.mla_decode_fp8_per_token(qn.cpu(), qs.cpu(), qr.cpu(), c["k_lora"].view(pages, 64, 1, 512),
                                                 c["k_scale"].view(pages, 64, 1, 1), c["k_rope"].view(pages, 64, 1, 64),
                                                 c["block_table"], c["cache_seqlens"], SCALE, True)
    emu, else_ = mla_ref.mla_decode_fp8_per_token_emulated(qn.cpu(), qs.cpu(), qr.cpu(), c["k_lora"].view(pages, 64, 1, 512),
                                                 c["k_scale"].view(pages, 64, 1, 1), c["k_rope"].view(pages, 64, 1, 64),
                                                 c["block_table"], c["cache_seqlens"], SCALE, True)
    ee = (o.cpu().double() - emu).abs()[0, 0]
    print(f"   vs emulation: rel={float(ee.mean()/emu.abs().mean()):.3e} max={float(ee.max()):.3e}; by row {[round(float(ee[r].mean()),4) for r in range(min(8,H))]}; worst idx {divmod(int(ee.argmax()), 512)}")
    err = (o.cpu().double() - ref).abs()[0, 0]      # [H, 512]
    print(f"lens={lens} H={H}: rel={float(err.mean()/ref.abs().mean()):.3e} lse_err={float((lse.cpu().double()-rlse).abs().max()):.3e}")
    print("  err by 64-col block:", [round(float(err[:, i*64:(i+1)*64].mean()), 4) for i in range(8)])
    print("  err by row (first 8):", [round(float(err[r].mean()), 4) for r in range(min(8, H))])
    print("  o[0,:4] ", o[0, 0, 0, :4].float().tolist(), " ref ", ref[0, 0, 0, :4].tolist())
