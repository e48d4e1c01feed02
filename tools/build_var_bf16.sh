#!/bin/bash
# Build A/B variants of the library from sed-edited copies of mla_decode_bf16.hip (never used by the product path):
#   tools/build_var_bf16.sh NAME 's/a/b/' [NAME2 's/c/d/' ...]   ->  fluent_mi355/libfluent_exp_bf16_NAME.so
set -e
cd /root/repo/sglang-fluentllm_amd/csrc
make -s > /dev/null 2>&1
while [ $# -ge 2 ]; do
  v=$1; e=$2; shift 2
  sed -e "$e" mla_decode_bf16.hip > /tmp/mla_decode_bf16_$v.hip
  if cmp -s mla_decode_bf16.hip /tmp/mla_decode_bf16_$v.hip && [ "$e" != "" ]; then echo "variant $v: sed expression changed nothing"; exit 1; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-inline-asm -c /tmp/mla_decode_bf16_$v.hip -o /tmp/mla_decode_bf16_$v.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../fluent_mi355/libfluent_exp_bf16_$v.so $(ls build/*.o | grep -v mla_decode_bf16.o) /tmp/mla_decode_bf16_$v.o
done
ls ../fluent_mi355/*.so
