#!/bin/bash
# GPU box: PMC passes over the K2-bf16 kernel at H = 128 and H = 16
cd /root/repo; mkdir -p gpurun_out
export CACHES=2 LAUNCHES=2 KREGEX=mla_decode_bf16 PROF_SCRIPT=tools/time_k2_bf16.py
PROF_ARGS="128 128 4096" timeout 600 bash tools/rocprof_pmc.sh gpurun_out/pmc_bf16_h128 > gpurun_out/pmc_bf16.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ["gpurun_out/pmc_bf16_h128"]:
    for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f)
        for k, v in agg.items(): print("  %-28s n=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
