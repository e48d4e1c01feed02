#!/bin/bash
# Build debug variants of the library around mla_decode_bf16.hip (never used by the product path): tools/build_exp_bf16.sh TIMING
set -e
cd /root/repo/sglang-fluentllm_amd/csrc
make -s > /dev/null 2>&1
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-inline-asm -DFL_MLA_$v -c mla_decode_bf16.hip -o /tmp/mla_decode_bf16_$v.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../fluent_mi355/libfluent_exp_bf16_$v.so $(ls build/*.o | grep -v mla_decode_bf16.o) /tmp/mla_decode_bf16_$v.o
done
ls ../fluent_mi355/*.so
