#!/bin/bash
# Build a variant of the round-6 grouped GEMM kernel into libfluent_exp_<name>.so: tools/build_gemm3_var.sh NAME "-DFL_G3_..." (schedule
# tables FL_G3_RD / FL_G3_DMA_E / FL_G3_DMA_O / FL_G3_CAP, -DFL_GEMM3_TIMING for the phase timer).  The other objects come from csrc/build.
set -e
cd /root/repo/sglang-fluentllm_amd/csrc
mkdir -p build/gexp
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -fno-slp-vectorize -Wno-inline-asm -Wno-unused-result"
/opt/rocm/bin/hipcc $FL $2 -c grouped_gemm_fp8_big3.hip -o build/gexp/big3_$1.o
OBJS=$(ls build/*.o | grep -v "grouped_gemm_fp8_big3")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../fluent_mi355/libfluent_exp_$1.so $OBJS build/gexp/big3_$1.o
G3FLAGS="$2" python /root/repo/tools/check_gemm3_isa.py
ls -la ../fluent_mi355/libfluent_exp_$1.so
