#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_mla_gpu.py -x -q -m gpu > gpurun_out/mla_tests.log 2>&1; echo "mla tests rc=$?" | tee -a gpurun_out/mla_tests.log; tail -3 gpurun_out/mla_tests.log
for a in "128 128 4096" "16 128 4096" "32 128 4096" "64 128 4096" "128 256 8192" "128 32 1024" "16 256 8192"; do timeout 300 python tools/time_k2_bf16.py $a 2>&1 | tail -1; done | tee gpurun_out/bf16_time.log
{ timeout 300 python tools/time_phases_bf16.py 128 128 4096; timeout 300 python tools/time_phases_bf16.py 16 128 4096; } 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/bf16_phases.log
