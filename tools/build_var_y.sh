#!/bin/bash
# Build A/B variants of the library from edited copies of mla_decode_fp8_y.hip (never used by the product path):
#   tools/build_var_y.sh NAME file.hip [NAME2 file2.hip ...]   ->  fluent_mi355/libfluent_exp_NAME.so   (file = a full replacement source)
set -e
cd /root/repo/sglang-fluentllm_amd/csrc
make -s > /dev/null 2>&1
while [ $# -ge 2 ]; do
  v=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -fno-slp-vectorize -Wno-inline-asm -c $f -o /tmp/mla_decode_fp8_y_$v.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../fluent_mi355/libfluent_exp_$v.so $(ls build/*.o | grep -v mla_decode_fp8_y.o) /tmp/mla_decode_fp8_y_$v.o
done
ls ../fluent_mi355/*.so
