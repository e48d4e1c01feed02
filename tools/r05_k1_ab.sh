# GPU box: round-5 K1 A/B (BASE = round-4 kernel) + the logit-spread cliff at bs=256 x 8192 + parity
set -x
mkdir -p gpurun_out/r05b
bash tools/ab_k1.sh 3 libfluent_exp_BASE.so libfluent_mi355.so > gpurun_out/r05b/ab_cfg2.txt 2>&1
for L in libfluent_exp_BASE.so libfluent_mi355.so; do for Q in 1 8 16; do
  QSCALE=$Q FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L FLUENT_MLA_LIB_TAG="$L q x$Q" LAYERS=8 python tools/time_k1.py 128 256 8192 2>/dev/null | tail -1
done; done > gpurun_out/r05b/cliff_256x8192.txt 2>&1
timeout 1500 python -m pytest tests/test_mla_gpu.py -x -q -m gpu > gpurun_out/r05b/pytest_mla.txt 2>&1
tail -5 gpurun_out/r05b/pytest_mla.txt
cat gpurun_out/r05b/ab_cfg2.txt gpurun_out/r05b/cliff_256x8192.txt
