#!/bin/bash
export DPQ_DEV=1
# four queries per pass at 125 M codes: in-tree library and every variant
mkdir -p gpurun_out
for lib in "" variants/lib_q4w6.so; do
for nq in 3 4; do
DPQ_LIB_PATH=${lib:+$PWD/$lib} DPQ_STREAM_MAX_QUERIES=8 timeout -k 10 400 python bench.py --codes 125000000 --data stream --queries $nq --steps 10 --warmup 2 --reps 3 --check 1 --no-cpu-baseline > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -5 gpurun_out/sm.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/sm.json").read().strip().splitlines()[-1])
print("lib '$lib' queries $nq:", round(d["value"],1), "q/s", round(d["ms_per_step"],3), "ms/step", "parity", d["parity_checked_queries"], flush=True)
PY
done; done 2>&1 | tee gpurun_out/stream_q4.txt
