set -x
mkdir -p gpurun_out/r05g
for r in 1 2 3; do for s in "16 128 4096" "16 256 8192" "32 128 4096"; do for L in libfluent_exp_NONT.so libfluent_mi355.so; do FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L FLUENT_MLA_LIB_TAG=$L LAYERS=61 python tools/time_k1.py $s 2>/dev/null | tail -1; done; done; done > gpurun_out/r05g/ab_nt.txt
cat gpurun_out/r05g/ab_nt.txt
timeout 600 python -m pytest tests/test_mla_gpu.py -x -q -m gpu -k "H16 or h16 or sq2 or small or decode" > gpurun_out/r05g/pytest.txt 2>&1; tail -3 gpurun_out/r05g/pytest.txt
