#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
D=$PWD/sglang-fluentllm_amd/fluent_mi355
{ for r in 1 2 3; do for L in ring2 ring3 ring3noslp ring2noslp; do
  for a in "128 128 4096" "64 128 4096" "16 128 4096"; do FLUENT_MLA_LIB_TAG=$L FLUENT_MI355_LIB=$D/libfluent_exp_bf16_$L.so timeout 300 python tools/time_k2_bf16.py $a 2>&1 | tail -1; done
done; done; } | tee gpurun_out/bf16_ab3.log
