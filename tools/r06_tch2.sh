#!/bin/bash
GT_LIB=libfluent_exp_TCHT.so GT_E=32 GT_ROWS=576 timeout 300 python tools/time_gemm3_phases.py 4096 7168 2>&1 | tail -6
GT_LIB=libfluent_exp_G3T.so GT_E=32 GT_ROWS=576 timeout 300 python tools/time_gemm3_phases.py 4096 7168 2>&1 | tail -6
