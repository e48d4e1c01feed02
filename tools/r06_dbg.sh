#!/bin/bash
mkdir -p gpurun_out/r06a
for L in NJB; do
(FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 120 python tools/r06_dbg.py 192 256 256 2>&1 | grep -v "^  File\|^Extension\|amdgpu.ids" | head -80)
done > gpurun_out/r06a/dbg.txt
cat gpurun_out/r06a/dbg.txt
