#!/bin/bash
# GPU box: K2-bf16 parity subset + timings
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mla_gpu.py -x -q -m gpu -k "bf16 or swap or trace or kvcache" > gpurun_out/bf16_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/bf16_tests.log
tail -5 gpurun_out/bf16_tests.log
for a in "128 128 4096" "16 128 4096" "64 128 4096" "128 256 8192" "128 32 1024"; do timeout 300 python tools/time_k2_bf16.py $a 2>&1 | tail -1; done | tee gpurun_out/bf16_time.log
