#!/bin/bash
# GPU box: A/B builds of the library on the SAME box on bench.py's cfg2_ragged variant (K1 only, graph replay), alternating.
# usage: tools/ab_ragged.sh <rounds> <libA.so> <libB.so> ...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L python -c "
import torch, bench
r = bench.k1_ragged_variant(torch.device('cuda:0'), int('${LAYERS:-61}'))
print('$L ragged', r['us_per_launch'], 'us', r['hbm_frac'])" 2>/dev/null | tail -1
  done
done
