import sys, os
sys.path.insert(0, "/root/repo/sglang-fluentllm_amd")
import torch, flash_mla_fp8 as fm
s = torch.full((128,), 4096, dtype=torch.int32, device="cuda")
m, ns = fm.get_mla_metadata(s, 128, 1)
print("X=", os.environ.get("FLUENT_MLA_X"), "num_parts", m.shape, "splits", int(ns[-1]))
