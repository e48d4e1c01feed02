set -x
mkdir -p gpurun_out/r05a
for s in "128 128 4096" "128 256 8192" "128 256 4096" "128 128 8192" "128 256 1024" "128 64 16384"; do LAYERS=4 timeout 300 python tools/time_k1.py $s; done > gpurun_out/r05a/time_k1.txt 2>&1
PROF_ARGS="2 2 128 256 8192" timeout 600 bash tools/rocprof_pmc.sh gpurun_out/r05a/pmc_256x8192 > gpurun_out/r05a/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/r05a/pmc_256x8192 > gpurun_out/r05a/pmc_256x8192_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05a/kt -o kt -- python tools/prof_mla.py 2 2 128 256 8192 > gpurun_out/r05a/kt.log 2>&1
find gpurun_out/r05a -name "*stats*" | head
