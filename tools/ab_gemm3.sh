#!/bin/bash
# GPU box: variant libraries of the round-6 grouped GEMM on the bench shapes, two rounds.  usage: tools/ab_gemm3.sh <T> <lib suffix> ...
T=$1; shift
for L in "$@" "$@"; do
  FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 600 python tools/bench_gemm.py $T 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'T', d['T'], 'w13', d['gate_up']['ms'], 'ms', d['gate_up']['TFLOPs'], 'TF | w2', d['down']['ms'], 'ms', d['down']['TFLOPs'], 'TF')"
done
