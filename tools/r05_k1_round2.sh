set -x
mkdir -p gpurun_out/r05d
timeout 1500 python -m pytest tests/test_mla_gpu.py tests/test_cfg5_gpu.py -x -q -m gpu > gpurun_out/r05d/pytest_mla.txt 2>&1; tail -3 gpurun_out/r05d/pytest_mla.txt
bash tools/ab_k1.sh 3 libfluent_exp_BASE.so libfluent_mi355.so libfluent_exp_NOSUMS.so > gpurun_out/r05d/ab_cfg2.txt 2>&1
for Q in 1 16 64; do QSCALE=$Q LAYERS=8 FLUENT_MLA_LIB_TAG="q x$Q" python tools/time_k1.py 128 256 8192 2>/dev/null | tail -1; done > gpurun_out/r05d/cliff.txt 2>&1
python bench.py --mode cfg4 --steps 5 --warmup 2 > gpurun_out/r05d/cfg4.json 2>gpurun_out/r05d/cfg4.err
cat gpurun_out/r05d/ab_cfg2.txt gpurun_out/r05d/cliff.txt; tail -c 1500 gpurun_out/r05d/cfg4.json
