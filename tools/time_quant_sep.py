"""GPU box: the reference call sequence's two quantise launches per layer (K5 quantize_and_cache_k + K4 quantize_ckv_per_token_head), one hipGraph of 61
pairs over distinct inputs, us per PAIR.  usage: python tools/time_quant_sep.py [bs] [H]   (FLUENT_MI355_LIB selects another build of the library)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import flash_mla_fp8 as fm
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = 61
slots = 128 * 64 * 70
g = torch.Generator(device=dev).manual_seed(0)
qs = [torch.randn(bs, H, 576, device=dev, generator=g).to(torch.bfloat16) for _ in range(8)]
ks = [torch.randn(bs, 1, 576, device=dev, generator=g).to(torch.bfloat16) for _ in range(8)]
loc = (torch.arange(bs, device=dev, dtype=torch.int32) * 4099 + 63)
k_lora = torch.zeros(slots, 1, 512, dtype=torch.uint8, device=dev)
k_scale = torch.zeros(slots, 1, 1, dtype=torch.float32, device=dev)
k_rope = torch.zeros(slots, 1, 64, dtype=torch.bfloat16, device=dev)
def pair(i):
    fm.quantize_and_cache_k(ks[i % 8], k_lora, k_scale, k_rope, loc, 512)
    return fm.quantize_ckv_per_token_head(qs[i % 8].view(bs, 1, H, 576), 512)
pair(0); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): pair(0)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for i in range(N): pair(i)
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): gr.replay()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"bs": bs, "H": H, "us_per_pair": round(e0.elapsed_time(e1) * 1e3 / (20 * N), 2), "tag": os.environ.get("FLUENT_MLA_LIB_TAG", "")}))
