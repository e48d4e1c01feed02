#!/bin/bash
# GPU box: the round's record — full GPU suite, randomized GEMM parity sweep, PMC passes of the 192 x 256 grouped GEMM, one-batch sweep, kernel trace of
# the bench, the bench line.  usage: bash tools/record.sh [skip-tests]   (output: gpurun_out/record/)
set -x
O=gpurun_out/record; mkdir -p $O
if [ "$1" != "skip-tests" ]; then
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
  timeout 900 python tools/stress_gemm.py 24 6 > $O/stress_gemm.txt 2>&1; tail -2 $O/stress_gemm.txt
fi
timeout 1500 bash tools/rocprof_pmc_gemm.sh $O/pmc_gemm > $O/pmc_gemm_summary.txt 2>&1; tail -30 $O/pmc_gemm_summary.txt
timeout 900 python tools/bench_one_batch.py --batch 1 16 128 --seq 4096 16384 > $O/bench_one_batch.txt 2>&1; tail -8 $O/bench_one_batch.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-gemm > $O/bench_profiled.json 2> $O/bench_profiled.err
DB=$(find $O/kt -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB > $O/kernel_trace_stats.txt 2>&1
head -12 $O/kernel_trace_stats.txt
rm -rf $O/kt
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
