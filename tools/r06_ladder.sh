#!/bin/bash
# GPU box, round 6: bounding ladder of the 192 x 256 grouped GEMM (big3): phase timer, MFMA slot stamps, garbage-result builds under the power sampler
mkdir -p gpurun_out/r06c
{
GT_E=32 GT_ROWS=576 timeout 300 python tools/time_gemm3_phases.py 4096 7168 2>&1 | tail -7
GT_E=32 GT_ROWS=576 timeout 300 python tools/time_gemm3_phases.py 7168 2048 2>&1 | tail -7
GT_E=32 GT_ROWS=576 timeout 300 python tools/time_gemm3_slots.py 4096 7168 2>&1 | tail -4
} | tee gpurun_out/r06c/phases_slots.txt
{
FLUENT_GEMM_BIG=2 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
FLUENT_GEMM_BIG=3 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
for L in NODMA NOBAR NORS NORD ALL4 "$@"; do
  FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
done
} | tee gpurun_out/r06c/ladder.txt
