#!/bin/bash
# GPU box: K1 as a function of query rows per request (s_q x H: MTP verify = 256-512 rows) at fixed KV bytes
cd /root/repo; mkdir -p gpurun_out
{ for a in "64 64 16384" "128 64 16384" "256 64 16384" "512 64 16384" "128 128 4096" "256 128 4096" "512 128 4096"; do LAYERS=4 timeout 600 python tools/time_k1.py $a 2>&1 | tail -1; done; } | tee gpurun_out/rows_sweep.txt
