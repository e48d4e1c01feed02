#!/bin/bash
# GPU box: same-box alternating A/B of K2-bf16 library variants (tools/build_var_bf16.sh): usage [SHAPES="128,128,4096 16,128,4096"] tools/r05_bf16_ab.sh <rounds> <variant> ...
cd /root/repo; mkdir -p gpurun_out
D=$PWD/sglang-fluentllm_amd/fluent_mi355
R=$1; shift
{ for r in $(seq $R); do for L in "$@"; do
  for a in ${SHAPES:-128,128,4096 64,128,4096 16,128,4096}; do a=${a//,/ }; FLUENT_MLA_LIB_TAG=$L FLUENT_MI355_LIB=$D/libfluent_exp_bf16_$L.so timeout 300 python tools/time_k2_bf16.py $a 2>&1 | tail -1; done
done; done; } | tee gpurun_out/bf16_ab.log | python -c "
import sys, json, collections
agg = collections.defaultdict(list)
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); agg[(d['tag'], d['H'], d['bs'], d['seq'])].append(d['us_per_launch'])
for k in sorted(agg): print(k, agg[k])"
