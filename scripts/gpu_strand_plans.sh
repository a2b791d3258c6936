#!/bin/bash
export DPQ_DEV=1
# level plans of the stream pass on the strand image, 125 M codes, one query
mkdir -p gpurun_out
for r in "" "8" "16" "4" "32"; do
DPQ_PLAN_RATIOS=$r timeout -k 10 500 python bench.py --codes 125000000 --data stream --queries ${Q:-1} --steps 10 --warmup 2 --reps 3 --check 1 --no-cpu-baseline --sustain-seconds 0 --host-steps 0 > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -5 gpurun_out/sm.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/sm.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("ratios '$r':", round(d["value"],1), "q/s", round(d["ms_per_step"],3), "ms/step launches", r["launches_per_step"], "cand/q", round(r["candidates_per_query"]), "parity", d["parity_checked_queries"], flush=True)
PY
done 2>&1 | tee gpurun_out/strand_plans.txt
