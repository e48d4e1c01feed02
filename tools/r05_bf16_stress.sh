#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mla_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -3
timeout 1200 python tools/stress_mla_bf16.py 60 5 2>&1 | grep -v Warning | tee gpurun_out/stress_bf16.txt | tail -8
