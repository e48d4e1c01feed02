#!/bin/bash
# GPU box: PMC passes (separate runs, no trace domains) over the compute-regime grouped GEMM (tools/prof_gemm.py: 32 experts x 512 rows, w13 and w2)
set -u
OUT=${1:-gpurun_out/pmc_gemm}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K="--kernel-include-regex grouped_gemm_fp8_big"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE $K --output-format csv -d $OUT/p1 -o p1 -- python tools/prof_gemm.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE FETCH_SIZE $K --output-format csv -d $OUT/p2 -o p2 -- python tools/prof_gemm.py > $OUT/p2.log 2>&1
# (a third pass with WRITE_SIZE / TCC counters hangs for > 40 minutes on this kernel: not run)
python tools/pmc_summary.py $OUT
