#!/bin/bash
# GPU box: small-batch decode step — sweep + kernel trace at bs = 1 and 16
cd /root/repo; O=gpurun_out/small; mkdir -p $O
python tools/bench_one_batch.py --batch 1 2 4 8 16 32 64 --seq 4096 2>/dev/null | tee $O/one_batch.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 1 16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt$b -o kt -- python tools/bench_one_batch.py --batch $b --seq 4096 --steps 3 > $O/kt$b.log 2>&1
  DB=$(find $O/kt$b -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB > $O/kt${b}_stats.txt 2>&1
  head -14 $O/kt${b}_stats.txt
done
