#!/bin/bash
mkdir -p gpurun_out/r06e
timeout 600 python -m pytest tests/test_mla_gpu.py -x -q -k "metadata" 2>&1 | tail -3 | tee gpurun_out/r06e/pytest_k3.txt
{ timeout 200 python tools/time_k3.py 2>&1 | tail -1; FLUENT_MLA_METADATA_TWO_WALKS=1 timeout 200 python tools/time_k3.py 2>&1 | tail -1; } | tee gpurun_out/r06e/time_k3.txt
