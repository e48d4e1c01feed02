#!/bin/bash
# GPU box: compare builds of the library on the SAME box, alternating (clock / box variation is several %).
# usage: tools/ab.sh <rounds> <libA.so> <libB.so> ...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['roofline']['us_per_launch'], 'us/launch', d['value'], 'tok/s')"
  done
done
