#!/bin/bash
# GPU box, round 6: state of the tree — gemm3 parity, big2 / big3 A/B on the bench shapes, the bench line
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gemm3_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/r06b/pytest_gemm3.txt
for T in 16384 32768; do
for B in 2 3 2 3; do
  FLUENT_GEMM_BIG=$B timeout 600 python tools/bench_gemm.py $T 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BIG=$B', 'T', d['T'], 'w13', d['gate_up']['ms'], 'ms', d['gate_up']['TFLOPs'], 'TF | w2', d['down']['ms'], 'ms', d['down']['TFLOPs'], 'TF | layer', d['moe_layer_ms(quant+gemm+silu+quant+gemm)'])"
done
done 2>&1 | tee gpurun_out/r06b/ab_big2_big3.txt
timeout 900 python bench.py > gpurun_out/r06b/bench.json 2> gpurun_out/r06b/bench.err; tail -c 6000 gpurun_out/r06b/bench.json
