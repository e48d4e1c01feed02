#!/bin/bash
# Build an experimental variant of the GEMM objects into libfluent_exp_<name>.so: tools/build_gemm_exp.sh NAME "-DFL_..."
set -e
cd /root/repo/sglang-fluentllm_amd/csrc
mkdir -p build/gexp
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -fno-slp-vectorize -Wno-inline-asm"
/opt/rocm/bin/hipcc $FL $2 -c grouped_gemm_fp8_big2.hip -o build/gexp/big2_$1.o &
/opt/rocm/bin/hipcc $FL $2 -c grouped_gemm_fp8.hip -o build/gexp/g_$1.o &
wait
OBJS=$(ls build/*.o | grep -v "grouped_gemm_fp8")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../fluent_mi355/libfluent_exp_$1.so $OBJS build/gexp/big2_$1.o build/gexp/g_$1.o
ls -la ../fluent_mi355/libfluent_exp_$1.so
