#!/bin/bash
# GPU box, round 6: filler-cost probe, skew variants of big3, name of the vendor fp8 kernel
mkdir -p gpurun_out/r06d
timeout 300 ./probes/probe_fillers 2>&1 | tee gpurun_out/r06d/probe_fillers.txt
{
for L in SK1 SK2 SK1ND; do
  FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
done
FLUENT_GEMM_BIG=3 timeout 120 python tools/power_gemm.py 3 2>/dev/null | tail -1
} | tee gpurun_out/r06d/skew.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/hb -o hb -- python $GRAFT_REPO_ROOT/tools/ref_hipblaslt_fp8.py > /tmp/hb.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/hb -name "*kernel_stats*" | head -1 | xargs -I{} cp {} gpurun_out/r06d/hipblaslt_kernel_stats.csv
head -12 gpurun_out/r06d/hipblaslt_kernel_stats.csv | cut -c1-400
tail -8 /tmp/hb.log
