#!/bin/bash
# GPU box: the compute-regime grouped GEMM (T = 16384 tokens top-8 over 256 experts, w13, random fp8 bytes: tools/power_gemm.py, 3 s loops) on the
# shipped library and on variant builds of the 192 x 256 kernel (tools/build_gemm3_var.sh NAME "-DFL_G3_..."; the experiment switches of round 6
# are probes/r06_gemm_big3_lab_switches.patch.txt), shipped first and last.  usage: tools/gemm_ladder.sh [NAME ...]   env FLUENT_G3_GRID with a
# -DFL_G3_GRIDENV build: number of workgroups.  Results of round 6: profiles/r06_gemm_big3_bounding_ladder.txt
mkdir -p gpurun_out/gemm_ladder
{
FLUENT_GEMM_BIG=2 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
FLUENT_GEMM_BIG=3 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
for L in "$@"; do
  FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
done
FLUENT_GEMM_BIG=3 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
} | tee -a gpurun_out/gemm_ladder/ladder.txt
