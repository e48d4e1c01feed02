#!/bin/bash
# GPU box: name of the fp8 kernel hipBLASLt selects for the dense problem of tools/ref_hipblaslt_fp8.py (kernel trace as csv)
mkdir -p gpurun_out/r06d
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hb -o hb -- python $GRAFT_REPO_ROOT/tools/ref_hipblaslt_fp8.py > /tmp/hb.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/hb -name "*stats*csv" | head -3
f=$(find /tmp/hb -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06d/hipblaslt_kernel_stats.csv
cut -c1-600 gpurun_out/r06d/hipblaslt_kernel_stats.csv | head -8
grep "^{" /tmp/hb.log | head
