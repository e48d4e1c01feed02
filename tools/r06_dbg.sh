#!/bin/bash
mkdir -p gpurun_out/r06a
(timeout 120 python tools/r06_dbg.py 192 256 256 2>&1 | grep -v "^  File\|^Extension" | head -60) > gpurun_out/r06a/dbg.txt
(timeout 120 python tools/r06_dbg.py 192,193,1,0,383 512 384 2>&1 | grep -v "^  File\|^Extension" | head -60) >> gpurun_out/r06a/dbg.txt
cat gpurun_out/r06a/dbg.txt
timeout 900 python -m pytest tests/test_gemm3_gpu.py -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -15 | tee gpurun_out/r06a/pytest_gemm3.txt
