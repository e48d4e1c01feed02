"""GPU box: BASELINE config 5 shape — one speculative-decode step of the MLA hot path captured in ONE hipGraph:
per layer, verify (s_q = 4 draft tokens, cache_seqlens = seq + 4) + 3 draft decode steps (s_q = 1, seq + 4 + i + 1),
each with its K5 store and K4 q-quantisation.  bs=64, seq=16384, per-token fp8 KV, H heads (default 64), L layers with
their own caches.  Prints one JSON line: ms per layer-step, achieved GB/s against the algorithmic bytes of the 4 calls."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, bench
import flash_mla_fp8 as fm

H = int(os.environ.get("H", 64)); BS = int(os.environ.get("BS", 64)); SEQ = int(os.environ.get("SEQ", 16384))
LAYERS = int(os.environ.get("LAYERS", 4)); DRAFT, STEPS = 4, 3
dev = torch.device("cuda:0")
wl = bench.build_workload(dev, LAYERS, BS, SEQ + 64, H, seed=7)       # one spare page per request for the new tokens
pages = wl["pages"]
g = torch.Generator(device=dev).manual_seed(1)
L = torch.full((BS,), SEQ, dtype=torch.int32, device=dev)
seq_v = L + DRAFT
seq_d = [L + DRAFT + i + 1 for i in range(STEPS)]
bt = wl["block_table"]
def locs(first, count):   # slots of tokens first .. first+count-1 of every request
    t = (first + torch.arange(count, device=dev)).view(1, -1).expand(BS, -1)
    return (bt.gather(1, (t // 64).long()).long() * 64 + t % 64).reshape(-1).to(torch.int32)
loc_v = locs(SEQ, DRAFT)
loc_d = [locs(SEQ + DRAFT + i, 1) for i in range(STEPS)]
qv = torch.randn(BS, DRAFT, H, 576, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
qd = [torch.randn(BS, 1, H, 576, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16) for _ in range(STEPS)]
kv = torch.randn(BS * DRAFT, 1, 576, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
kd = [torch.randn(BS, 1, 576, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16) for _ in range(STEPS)]
meta_v, ns_v = fm.get_mla_metadata(seq_v, DRAFT * H, 1)
meta_d, ns_d = zip(*[fm.get_mla_metadata(seq_d[i], H, 1) for i in range(STEPS)])

def attend(l, q, seqlens, meta, ns):
    k_lora, k_scale, k_rope = wl["caches"][l]
    qn, qs, qr = fm.quantize_ckv_per_token_head(q, 512)
    return fm.flash_mla_ckv_fp8_per_token(qn, qr, k_lora.view(pages, 64, 1, 512), k_rope.view(pages, 64, 1, 64), qs,
                                          k_scale.view(pages, 64, 1, 1), bt, seqlens, 512, meta, ns, bench.SCALE, True)

def step():
    for l in range(LAYERS):
        k_lora, k_scale, k_rope = wl["caches"][l]
        fm.quantize_and_cache_k(kv, k_lora, k_scale, k_rope, loc_v, 512)
        attend(l, qv, seq_v, meta_v, ns_v)
        for i in range(STEPS):
            fm.quantize_and_cache_k(kd[i], k_lora, k_scale, k_rope, loc_d[i], 512)
            attend(l, qd[i], seq_d[i], meta_d[i], ns_d[i])

step(); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
graph.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps): graph.replay()
e1.record(); torch.cuda.synchronize()
ms_layer = e0.elapsed_time(e1) / (reps * LAYERS)
alg = bench.algorithmic_bytes(BS, SEQ + DRAFT, H, DRAFT) + sum(bench.algorithmic_bytes(BS, SEQ + DRAFT + i + 1, H, 1) for i in range(STEPS))
print(json.dumps({"workload": f"MTP step in one hipGraph: verify s_q=4 + 3 draft decodes, bs={BS} seq={SEQ} H={H}, {LAYERS} layers",
                  "ms_per_layer_step": round(ms_layer, 4), "algorithmic_GB_per_layer_step": round(alg / 1e9, 3),
                  "GBs": round(alg / ms_layer / 1e6, 1), "hbm_frac": round(alg / ms_layer / 1e6 / 8000, 4)}))
