set -x
mkdir -p gpurun_out/r05i
for r in 1 2 3; do for s in "128 128 4096" "128 256 8192" "64 128 4096"; do for L in libfluent_exp_NT1ONLY.so libfluent_mi355.so; do FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/$L FLUENT_MLA_LIB_TAG=$L LAYERS=61 python tools/time_k1.py $s 2>/dev/null | tail -1; done; done; done > gpurun_out/r05i/ab_nt3.txt
cat gpurun_out/r05i/ab_nt3.txt
bash tools/rocprof_pmc.sh gpurun_out/r05i/pmc_cfg2 > gpurun_out/r05i/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/r05i/pmc_cfg2 > gpurun_out/r05i/pmc_cfg2_summary.txt 2>&1
PROF_ARGS="2 2 16 128 4096" bash tools/rocprof_pmc.sh gpurun_out/r05i/pmc_h16 > gpurun_out/r05i/pmc_h16.log 2>&1
python tools/pmc_summary.py gpurun_out/r05i/pmc_h16 > gpurun_out/r05i/pmc_h16_summary.txt 2>&1
grep -E "FETCH|WRITE|TCC|GRBM|MFMA_BUSY|INSTS_VALU|INSTS_MFMA" gpurun_out/r05i/pmc_cfg2_summary.txt gpurun_out/r05i/pmc_h16_summary.txt
