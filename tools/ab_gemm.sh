#!/bin/bash
# GPU box: A/B builds of the library on the grouped GEMM bench, same box.  usage: tools/ab_gemm.sh <T> <libA> <libB> ...
T=$1; shift
for L in "$@" "$@"; do
  FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L python tools/bench_gemm.py $T 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'T', d['T'], 'w13', d['gate_up']['TFLOPs'], 'TF', d['gate_up']['GBs'], 'GB/s | w2', d['down']['TFLOPs'], 'TF', d['down']['GBs'], 'GB/s')"
done
