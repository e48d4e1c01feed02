set -x
mkdir -p gpurun_out/r05h
for r in 1 2 3; do
  for L in libfluent_exp_NONT.so libfluent_mi355.so; do FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L FLUENT_MLA_LIB_TAG=$L LAYERS=61 python tools/time_k1.py 16 128 4096 2>/dev/null | tail -1; done
  for s in "128 128 4096" "64 128 4096"; do for L in libfluent_mi355.so libfluent_exp_NTALL.so; do FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L FLUENT_MLA_LIB_TAG=$L LAYERS=61 python tools/time_k1.py $s 2>/dev/null | tail -1; done; done
done > gpurun_out/r05h/ab_nt2.txt
cat gpurun_out/r05h/ab_nt2.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r05h/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r05h/pytest_gpu.txt
