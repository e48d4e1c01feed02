#!/bin/bash
for L in "$@"; do
  if [ "$L" = "big2" ]; then FLUENT_GEMM_BIG=2 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
  else FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1; fi
done
