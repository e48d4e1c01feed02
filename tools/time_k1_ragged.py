import sys, os, json, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "sglang-fluentllm_amd"))
import bench
print(json.dumps(bench.k1_ragged_variant(torch.device("cuda:0"))))
