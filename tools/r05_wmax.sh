#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mla_gpu.py tests/test_absorb_gpu.py -x -q -m gpu -k "quant or cache_k or dequant or absorb or golden" 2>&1 | tail -2
D=$PWD/sglang-fluentllm_amd/fluent_mi355
for r in 1 2 3; do for L in wmaxold wmaxnew; do for a in "128 128" "16 128" "1 128" "128 16"; do FLUENT_MLA_LIB_TAG=$L FLUENT_MI355_LIB=$D/libfluent_exp_$L.so python tools/time_quant_sep.py $a 2>/dev/null | tail -1; done; done; done | python -c "
import sys, json, collections
agg = collections.defaultdict(list)
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); agg[(d['tag'], d['bs'], d['H'])].append(d['us_per_pair'])
for k in sorted(agg): print(k, agg[k])"
