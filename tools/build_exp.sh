#!/bin/bash
# Build experimental / debug variants of the library (never used by the product path): tools/build_exp.sh NOCOMPUTE NODMA ...
set -e
cd /root/repo/sglang-fluentllm_amd/csrc
mkdir -p build/exp
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include"
for f in fl_common mla_quant mla_metadata mla_decode_bf16 grouped_gemm_fp8 grouped_gemm_fp8_big2 quant_act norm_fused ep_a2a moe_gate rope kv_move ep_scatter_gather comm_oneshot bmm_bf16 mla_absorb; do
  [ build/exp/$f.o -nt $f.hip ] || /opt/rocm/bin/hipcc $FL -c $f.hip -o build/exp/$f.o 2>/dev/null
done
for v in "$@"; do
  EXTRA_V=""; case $v in TIMING_*) EXTRA_V="-DFL_MLA_TIMING -DFL_Y_${v#TIMING_}";; V_*) EXTRA_V="-DFL_Y_${v#V_}";; esac
  /opt/rocm/bin/hipcc $FL -DFL_EXP_$v -DFL_MLA_$v $EXTRA_DEFS -mllvm -amdgpu-mfma-vgpr-form=1 -c mla_decode_fp8.hip -o build/exp/mla_decode_fp8_$v.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc $FL -fno-slp-vectorize -Wno-inline-asm -DFL_EXP_$v -DFL_MLA_$v $EXTRA_DEFS $EXTRA_V -c mla_decode_fp8_y.hip -o build/exp/mla_decode_fp8_y_$v.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../fluent_mi355/libfluent_exp_$v.so build/exp/mla_decode_fp8_y_$v.o build/exp/fl_common.o build/exp/mla_quant.o build/exp/mla_metadata.o build/exp/mla_decode_bf16.o build/exp/grouped_gemm_fp8.o build/exp/grouped_gemm_fp8_big2.o build/exp/quant_act.o build/exp/norm_fused.o build/exp/ep_a2a.o build/exp/moe_gate.o build/exp/rope.o build/exp/kv_move.o build/exp/ep_scatter_gather.o build/exp/comm_oneshot.o build/exp/bmm_bf16.o build/exp/mla_absorb.o build/exp/mla_decode_fp8_$v.o
done
ls -la ../fluent_mi355/*.so
