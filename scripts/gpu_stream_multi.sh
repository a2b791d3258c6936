#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# Q queries per pass (stream_kernel<M, Q>) against the 64-query filter path: where the switch-over sits
mkdir -p gpurun_out
CODES=${CODES:-125000000}
for cfg in "1 8" "2 8" "4 8" "4 0" "8 8" "8 0" "16 16" "16 0" "32 32" "32 0"; do
set -- $cfg
DPQ_STREAM_MAX_QUERIES=$2 timeout -k 10 400 python bench.py --codes $CODES --data stream --queries $1 --steps 10 --warmup 2 --reps 3 --check 1 --no-cpu-baseline > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -5 gpurun_out/sm.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/sm.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("codes $CODES queries $1 stream_max $2:", round(d["value"],1), "q/s", round(d["ms_per_step"],3), "ms/step scan", round(r["scan_ms_per_step"],3), "launches", r["launches_per_step"], "algorithmic GB/s", round(r["algorithmic_hbm"]["GBps"],1), "parity", d["parity_checked_queries"], flush=True)
PY
done 2>&1 | tee gpurun_out/stream_multi_$CODES.txt
