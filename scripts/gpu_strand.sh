#!/bin/bash
export DPQ_DEV=1
# the stream pass on the strand image (lane per run) against the wavefront-per-chunk decode, 125 M codes
mkdir -p gpurun_out
CODES=${CODES:-125000000}
for cfg in "1 1" "1 0" "2 1" "4 1" "4 0"; do
set -- $cfg
DPQ_STRANDS=$2 timeout -k 10 500 python bench.py --codes $CODES --data stream --queries $1 --steps 10 --warmup 2 --reps 3 --check 1 --no-cpu-baseline > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -5 gpurun_out/sm.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/sm.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("codes $CODES queries $1 strands $2:", round(d["value"],1), "q/s", round(d["ms_per_step"],3), "ms/step scan", round(r["scan_ms_per_step"],3), "algorithmic GB/s", round(r["algorithmic_hbm"]["GBps"],1), "parity", d["parity_checked_queries"], flush=True)
PY
done 2>&1 | tee gpurun_out/strand_$CODES.txt
