"""GPU box: is K1 (bs=128 seq=4096 H=128) power-limited?  Replays the 61-layer K1 graph for a few seconds per variant while `rocm-smi`
samples socket power and shader clock: (a) the bench's N(0,1) cache, (b) every request reading the same pages (L2 hits), (c) an all-zero
cache and query (no data toggling).  usage: python tools/k1_power_probe.py [H bs seq]"""
import json, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_fp8 as fm

H = int(sys.argv[1]) if len(sys.argv) > 1 else bench.H
bs = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BS
seq = int(sys.argv[3]) if len(sys.argv) > 3 else bench.SEQ
layers = int(os.environ.get("LAYERS", "24"))
dev = torch.device("cuda:0")


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power[^:]*:\s*([0-9.]+)", r)
            sc = re.findall(r"sclk clock level[^(]*\((\d+)Mhz\)", r)
            mc = re.findall(r"mclk clock level[^(]*\((\d+)Mhz\)", r)
            out.append((float(pw[0]) if pw else None, int(sc[0]) if sc else None, int(mc[0]) if mc else None))
        except Exception:   # noqa
            out.append((None, None, None))
        time.sleep(0.1)


def run(tag, wl):
    meta, ns = fm.get_mla_metadata(wl["seqlens"], H, 1)
    qn, qs, qr = fm.quantize_ckv_per_token_head(wl["q"], 512)
    pages = wl["pages"]

    def k1(l):
        k_lora, k_scale, k_rope = wl["caches"][l]
        fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                       k_scale.view(pages, 64, 1, 1), wl["block_table"], wl["seqlens"], 512, meta, ns, bench.SCALE, True)
    for l in range(layers): k1(l)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for l in range(layers): k1(l)
    g.replay(); torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples)); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(20): g.replay()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / (n * layers)
    pw = [p for p, _, _ in samples if p]; sc = [s for _, s, _ in samples if s]; mc = [m for _, _, m in samples if m]
    print(json.dumps({"variant": tag, "us_per_launch": round(us, 1), "power_W_mean": round(sum(pw) / max(len(pw), 1), 1),
                      "power_W_max": max(pw) if pw else None, "sclk_MHz_mean": round(sum(sc) / max(len(sc), 1)) if sc else None,
                      "mclk_MHz_mean": round(sum(mc) / max(len(mc), 1)) if mc else None, "samples": len(samples)}), flush=True)


wl = bench.build_workload(dev, layers, bs, seq, H, seed=1)
run("N(0,1) cache, HBM-cold pages", wl)
bt = wl["block_table"]
wl["block_table"] = bt[torch.zeros(bs, dtype=torch.long, device=dev)].contiguous()
run("every request reads request 0's pages (L2 hits)", wl)
wl["block_table"] = bt
for c in wl["caches"]:
    c[0].zero_(); c[2].zero_()
wl["q"] = torch.zeros_like(wl["q"])
run("all-zero cache and query", wl)
