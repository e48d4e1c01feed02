set -x
mkdir -p gpurun_out/r05c
for L in libfluent_exp_BASETIMING.so libfluent_exp_TIMING.so libfluent_exp_BASETIMING.so libfluent_exp_TIMING.so; do echo "== $L"; TIMING_LIB=$L LAYERS=24 python tools/time_phases_y.py 128 128 4096 2>/dev/null; done > gpurun_out/r05c/phases.txt 2>&1
python tools/k1_power_probe.py > gpurun_out/r05c/k1_power.txt 2>&1
cat gpurun_out/r05c/k1_power.txt
