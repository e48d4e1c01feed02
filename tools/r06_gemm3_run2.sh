#!/bin/bash
mkdir -p gpurun_out/r06a
timeout 600 python -m pytest tests/test_gemm3_gpu.py -x -q 2>&1 | tail -2
bash tools/r06_gemm3_ab.sh 16384
GT_E=32 GT_ROWS=576 timeout 300 python tools/time_gemm3_phases.py 4096 7168 2>&1 | tail -7 | tee gpurun_out/r06a/phases_w13.txt
GT_E=32 GT_ROWS=576 timeout 300 python tools/time_gemm3_phases.py 7168 2048 2>&1 | tail -7 | tee gpurun_out/r06a/phases_w2.txt
