"""GPU box: launch the same small MLA decode many times and compare every output bit for bit with the first one (a race in
the loader / wait protocol would show as a mismatch).  usage: tools/determinism_mla.py [iterations]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_fp8 as fm
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
res = {}
for bs, seq, H in ((1, 128, 16), (2, 129, 16), (1, 64, 128), (3, 200, 64), (1, 9000, 128), (3, 2000, 64), (5, 3000, 128)):   # the last three: split requests, merged in-kernel by their first piece — the SAME metadata (merge counters) serves every launch
    wl = bench.build_workload(dev, 1, bs, seq, H, seed=bs + seq)
    meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
    qn, qs, qr = fm.quantize_ckv_per_token_head(wl["q"], 512)
    k_lora, k_scale, k_rope = wl["caches"][0]; pages = wl["pages"]
    def run():
        return fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                              k_scale.view(pages, 64, 1, 1), wl["block_table"], wl["seqlens"], 512, meta, ns,
                                              bench.SCALE, True)
    o0, l0 = run(); o0, l0 = o0.clone(), l0.clone()
    bad = 0
    for i in range(N):
        if i % 7 == 0:   # disturb timing: a competing copy on another stream
            with torch.cuda.stream(torch.cuda.Stream()):
                torch.empty(1 << 24, device=dev).fill_(1.0)
        o, l = run()
        bad += int(not (torch.equal(o.view(torch.int16), o0.view(torch.int16)) and torch.equal(l, l0)))
    res[f"bs{bs}_seq{seq}_H{H}"] = bad
print(json.dumps({"iterations": N, "mismatching_launches": res}))
