#!/bin/bash
mkdir -p gpurun_out/r06c
L=${1:-TCH}
FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 900 python -m pytest tests/test_gemm3_gpu.py -x -q 2>&1 | tail -3
bash tools/r06_ladder2.sh "$@"
