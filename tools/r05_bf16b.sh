#!/bin/bash
# GPU box: K2-bf16 A/B of the non-temporal page stream + an L2-resident run (memory latency vs the rest of the step)
cd /root/repo; mkdir -p gpurun_out
D=$PWD/sglang-fluentllm_amd/fluent_mi355
{
for r in 1 2; do
for L in libfluent_exp_bf16_ntoff.so libfluent_exp_bf16_nton.so; do
  for a in "128 128 4096" "16 128 4096" "64 128 4096" "32 128 4096"; do FLUENT_MLA_LIB_TAG=$L FLUENT_MI355_LIB=$D/$L timeout 300 python tools/time_k2_bf16.py $a 2>&1 | tail -1; done
done; done
for a in "128 128 4096" "16 128 4096" "64 128 4096"; do SMALLSET=64 FLUENT_MLA_LIB_TAG=smallset64 timeout 300 python tools/time_k2_bf16.py $a 2>&1 | tail -1; done
} | tee gpurun_out/bf16_ab.log
