#!/bin/bash
# GPU box, round 6: A/B of the grouped GEMM kernels on the bench shapes (FLUENT_GEMM_BIG=2: 256 x 256 8-wave kernel, 3: 192 x 256 4-wave kernel)
mkdir -p gpurun_out/r06a
for B in 2 3 2 3; do
  FLUENT_GEMM_BIG=$B timeout 600 python tools/bench_gemm.py ${1:-16384} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BIG=$B', 'T', d['T'], 'w13', d['gate_up']['ms'], 'ms', d['gate_up']['TFLOPs'], 'TF | w2', d['down']['ms'], 'ms', d['down']['TFLOPs'], 'TF | layer', d['moe_layer_ms(quant+gemm+silu+quant+gemm)'])"
done 2>&1 | tee -a gpurun_out/r06a/ab_big2_big3.txt
