#!/bin/bash
mkdir -p gpurun_out/r06c
for G in 256 192 128 64 32 8; do
  FLUENT_G3_GRID=$G FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_GRID.so timeout 200 python tools/power_gemm.py 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $G', d['ms'], 'ms', d['TFLOPs'], 'TF', round(d['TFLOPs']/$G,2), 'TF per CU')"
done | tee gpurun_out/r06c/grid.txt
