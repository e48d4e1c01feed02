"""GPU box: a bench_one_batch-style decode sweep of the MLA hot path (the reference's harness is
python/sglang/bench_one_batch.py:309-405: per batch size / input length, N decode steps, MEDIAN step latency -> tokens/s).
One step = K3 once + 61 layers x (K5 store, K4 quantise q, K1 decode) in one hipGraph, synthetic DeepSeek-V3 shapes, per-token
fp8 KV.  Layers above what fits in HBM are emulated by replaying LAYERS_RESIDENT distinct layer caches round-robin (stated
in the output).  usage: tools/bench_one_batch.py [--batch 1 16 64 128 256] [--seq 1024 4096 16384] [--heads 128] [--steps 20]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_fp8 as fm

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[1, 16, 64, 128, 256])
ap.add_argument("--seq", type=int, nargs="+", default=[1024, 4096, 16384])
ap.add_argument("--heads", type=int, default=128)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
LAYERS = bench.LAYERS
for bs in a.batch:
    for seq in a.seq:
        per_layer = (bs * ((seq + 63) // 64) + 1) * 64 * 644
        resident = max(1, min(LAYERS, int(40e9 // per_layer)))          # distinct layer caches kept in HBM
        if per_layer > 60e9:
            continue
        wl = bench.build_workload(dev, resident, bs, seq, a.heads, seed=bs + seq)

        def step():
            meta, ns = fm.get_mla_metadata(wl["seqlens"], a.heads, 1)
            for l in range(LAYERS):
                bench.layer_call(fm, wl, l % resident, meta, ns)

        step(); torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): step()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr): step()
        gr.replay(); torch.cuda.synchronize()
        lat = []
        for _ in range(a.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            lat.append(e0.elapsed_time(e1))
        med = sorted(lat)[len(lat) // 2]
        gb = bench.algorithmic_bytes(bs, seq, a.heads, 1) * LAYERS / 1e9
        print(json.dumps({"batch": bs, "seq": seq, "heads": a.heads, "median_step_ms": round(med, 3),
                          "decode_tokens_per_s": round(bs / med * 1e3, 1), "GBs": round(gb / med * 1e3, 1),
                          "hbm_frac": round(gb / med / 8.0, 4), "distinct_layer_caches": resident}), flush=True)
        del wl, gr
        torch.cuda.empty_cache()
