"""GPU box: the fused K5 + K4 launch (fl_mla_quant_q_store_k) at the bench shape: bs new K rows + bs*H query rows, hipGraph of 8 calls over
distinct inputs.  usage: python tools/time_quant.py [bs] [H]   (FLUENT_MI355_LIB selects another build of the library)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import flash_mla_fp8 as fm
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
slots = 128 * 64 * 70
g = torch.Generator(device=dev).manual_seed(0)
qs = [torch.randn(bs, H, 576, device=dev, generator=g).to(torch.bfloat16) for _ in range(8)]
ks = [torch.randn(bs, 576, device=dev, generator=g).to(torch.bfloat16) for _ in range(8)]
loc = (torch.arange(bs, device=dev, dtype=torch.int32) * 4099 + 63)
k_lora = torch.zeros(slots, 512, dtype=torch.uint8, device=dev).view(torch.float8_e4m3fn)
k_scale = torch.zeros(slots, dtype=torch.float32, device=dev)
k_rope = torch.zeros(slots, 64, dtype=torch.bfloat16, device=dev)
def call(i):
    return fm.quantize_q_and_cache_k(qs[i], ks[i], k_lora, k_scale, k_rope, loc, 512)
call(0); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    call(0)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for i in range(8): call(i)
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): gr.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 400
byt = bs * H * (576 * 2 + 512 + 128 + 4) + bs * (576 * 2 + 644)
print(json.dumps({"bs": bs, "H": H, "us": round(us, 2), "GBs": round(byt / us / 1e3, 1)}))
