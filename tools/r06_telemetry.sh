#!/bin/bash
# GPU box: raw telemetry (amd-smi / rocm-smi) while a GEMM variant loops: which limiter pulls the shader clock down?
mkdir -p gpurun_out/tele
L=${1:-BSH}
FLUENT_MI355_LIB=$PWD/sglang-fluentllm_amd/fluent_mi355/libfluent_exp_$L.so timeout 200 python tools/power_gemm.py 25 > gpurun_out/tele/loop_$L.txt 2>&1 &
PID=$!
sleep 75   # torch import + operand generation on a fresh box
for i in 1 2 3; do
  echo "=== sample $i $(date +%s)" >> gpurun_out/tele/smi_$L.txt
  timeout 20 rocm-smi --showpower --showclocks --showtemp --showperflevel --showvoltage 2>&1 | grep -v "^$\|====" >> gpurun_out/tele/smi_$L.txt
  timeout 20 amd-smi metric -g 0 --power --clock --temperature --throttle --usage 2>&1 | head -150 >> gpurun_out/tele/amdsmi_$L.txt
  sleep 2
done
wait $PID
cat gpurun_out/tele/loop_$L.txt | tail -2
