set -x
mkdir -p gpurun_out/r05k
for L in libfluent_exp_GEMMOLD.so libfluent_exp_NT32.so libfluent_exp_GEMMOLD.so libfluent_exp_NT32.so; do echo "## $L"; FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L python tools/bench_gemm.py 128 256 512 1024 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['T'], d['rows_per_expert'], 'w13', d['gate_up']['GBs'], 'w2', d['down']['GBs'], 'layer_ms', d['moe_layer_ms(quant+gemm+silu+quant+gemm)'])
"; done > gpurun_out/r05k/gemm_nt_sweep.txt
cat gpurun_out/r05k/gemm_nt_sweep.txt
for L in libfluent_exp_GEMMOLD.so libfluent_exp_NT32.so; do echo "## $L"; FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L python tools/bench_dense.py 2>/dev/null | tail -20; done > gpurun_out/r05k/dense_nt.txt
cat gpurun_out/r05k/dense_nt.txt
