#!/bin/bash
# GPU box: randomized parity sweeps of the final round-5 kernels (K1 dispatch, grouped / dense GEMM, fused query side)
cd /root/repo; O=gpurun_out/r05_stress; mkdir -p $O
timeout 1500 python tools/stress_mla.py 60 11 2>&1 | grep -v Warning > $O/stress_mla.txt; tail -2 $O/stress_mla.txt
timeout 1500 python tools/stress_gemm.py 2>&1 | grep -v Warning > $O/stress_gemm.txt; tail -2 $O/stress_gemm.txt
timeout 900 python tools/stress_absorb.py 2>&1 | grep -v Warning > $O/stress_absorb.txt; tail -2 $O/stress_absorb.txt
BESIDE_GEMM=1 timeout 900 python tools/determinism_ragged.py 2>&1 | grep -v Warning > $O/determinism.txt; tail -3 $O/determinism.txt
