set -x
mkdir -p gpurun_out/r05j
for r in 1 2 3; do for s in "128 128 4096" "128 256 8192" "64 128 4096" "16 128 4096"; do for L in libfluent_exp_NT1ONLY.so libfluent_mi355.so; do FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L FLUENT_MLA_LIB_TAG=$L LAYERS=61 python tools/time_k1.py $s 2>/dev/null | tail -1; done; done; done > gpurun_out/r05j/ab_nt4.txt
cat gpurun_out/r05j/ab_nt4.txt
bash tools/rocprof_pmc.sh gpurun_out/r05j/pmc_cfg2 > gpurun_out/r05j/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/r05j/pmc_cfg2 > gpurun_out/r05j/pmc_cfg2_summary.txt 2>&1
grep -E "FETCH|WRITE|TCC|GRBM" gpurun_out/r05j/pmc_cfg2_summary.txt
python tools/bench_gemm.py > gpurun_out/r05j/bench_gemm_nt.txt 2>&1; tail -30 gpurun_out/r05j/bench_gemm_nt.txt
