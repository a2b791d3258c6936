#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# one query per pass (stream_kernel) against the 64-query filter path on big and small shards
mkdir -p gpurun_out
for cfg in "125000000 1 2" "125000000 1 0" "125000000 2 2" "125000000 4 4" "125000000 4 0" "12500000 1 2" "12500000 1 0"; do
set -- $cfg
DPQ_STREAM_MAX_QUERIES=$3 timeout -k 10 400 python bench.py --codes $1 --data stream --queries $2 --steps 10 --warmup 2 --reps 3 --check 1 --no-cpu-baseline > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -5 gpurun_out/sm.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/sm.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("codes $1 queries $2 stream_max $3:", round(d["value"],1), "q/s", round(d["ms_per_step"],3), "ms/step scan", round(r["scan_ms_per_step"],3), "launches", r["launches_per_step"], "algorithmic GB/s", round(r["algorithmic_hbm"]["GBps"],1), "parity", d["parity_checked_queries"], flush=True)
PY
done 2>&1 | tee gpurun_out/stream_mode.txt
