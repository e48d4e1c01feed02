#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/r05_bf16.sh
{ timeout 300 python tools/time_phases_bf16.py 128 128 4096; timeout 300 python tools/time_phases_bf16.py 16 128 4096; } 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/bf16_phases.log
