# GPU box: the round's record — full GPU suite, PMC passes of the final K1 (cfg2 and the TP8 shard), kernel trace of the bench, the bench line
set -x
O=gpurun_out/r05_record; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
bash tools/rocprof_pmc.sh $O/pmc_cfg2 > $O/pmc_cfg2.log 2>&1
python tools/pmc_summary.py $O/pmc_cfg2 > $O/pmc_cfg2_summary.txt 2>&1
PROF_ARGS="2 2 16 128 4096" bash tools/rocprof_pmc.sh $O/pmc_h16 > $O/pmc_h16.log 2>&1
python tools/pmc_summary.py $O/pmc_h16 > $O/pmc_h16_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-gemm > $O/bench_profiled.json 2> $O/bench_profiled.err
DB=$(find $O/kt -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB > $O/kernel_trace_stats.txt 2>&1
head -12 $O/kernel_trace_stats.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
CACHES=2 LAUNCHES=2 KREGEX=mla_decode_bf16 PROF_SCRIPT=tools/time_k2_bf16.py PROF_ARGS="128 128 4096" bash tools/rocprof_pmc.sh $O/pmc_bf16 > $O/pmc_bf16.log 2>&1
KREGEX=mla_decode_bf16 python tools/pmc_summary.py $O/pmc_bf16 > $O/pmc_bf16_summary.txt 2>&1
