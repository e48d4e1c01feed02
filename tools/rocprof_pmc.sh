#!/bin/bash
# usage: [PROF_ARGS="layers reps H bs seq"] tools/rocprof_pmc.sh <outdir> -- [PROF_SCRIPT=tools/x.py KREGEX=kernel] runs three separate --pmc passes (no trace domains combined) on tools/prof_mla.py
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K="--kernel-include-regex ${KREGEX:-mla_decode_y}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE $K --output-format csv -d $OUT/p1 -o p1 -- python ${PROF_SCRIPT:-tools/prof_mla.py} ${PROF_ARGS:-2 2} > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE FETCH_SIZE $K --output-format csv -d $OUT/p2 -o p2 -- python ${PROF_SCRIPT:-tools/prof_mla.py} ${PROF_ARGS:-2 2} > $OUT/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM $K --output-format csv -d $OUT/p3 -o p3 -- python ${PROF_SCRIPT:-tools/prof_mla.py} ${PROF_ARGS:-2 2} > $OUT/p3.log 2>&1
find $OUT -name "*.csv" | head
