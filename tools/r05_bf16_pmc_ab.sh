#!/bin/bash
# GPU box: K2-bf16 parity subset, LDS bank-conflict counters and timing after a layout change
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mla_gpu.py -x -q -m gpu -k "bf16 or swap or trace or kvcache" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CACHES=2 LAUNCHES=2 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-include-regex mla_decode_bf16 --output-format csv -d gpurun_out/pmc_bf16_x -o p -- python tools/time_k2_bf16.py 128 128 4096 > gpurun_out/pmc_bf16_x.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bf16_x
for a in "128 128 4096" "64 128 4096" "16 128 4096"; do timeout 300 python tools/time_k2_bf16.py $a 2>&1 | tail -1; done
