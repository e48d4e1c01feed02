#!/bin/bash
# GPU box, round 6: first light of the 192 x 256 one-wave-per-SIMD grouped GEMM — parity, then A/B against big2 on the bench shapes.
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_gemm3_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r06a/pytest_gemm3.txt
cat gpurun_out/r06a/pytest_gemm3.txt
for B in 2 3 2 3; do
  FLUENT_GEMM_BIG=$B timeout 600 python tools/bench_gemm.py 16384 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BIG=$B', 'w13', d['gate_up']['ms'], 'ms', d['gate_up']['TFLOPs'], 'TF | w2', d['down']['ms'], 'ms', d['down']['TFLOPs'], 'TF | layer', d['moe_layer_ms(quant+gemm+silu+quant+gemm)'])"
done 2>&1 | tee gpurun_out/r06a/ab_big2_big3.txt
