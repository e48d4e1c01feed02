#!/bin/bash
mkdir -p gpurun_out/r06c
{
FLUENT_GEMM_BIG=3 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
for L in "$@"; do
  FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
done
FLUENT_GEMM_BIG=3 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
} | tee -a gpurun_out/r06c/ladder2.txt
