#!/bin/bash
# GPU box: A/B builds of the library on the SAME box with tools/time_k1.py (K1 only, graph replay), alternating.
# usage: tools/ab_k1.sh <rounds> <libA.so> <libB.so> ...     (env H, BS, SEQ, SHARE_PAGES pass through)
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L FLUENT_MLA_LIB_TAG=$L LAYERS=${LAYERS:-61} python tools/time_k1.py ${H:-128} ${BS:-128} ${SEQ:-4096} 2>/dev/null | tail -1
  done
done
