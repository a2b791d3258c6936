#!/bin/bash
export DPQ_DEV=1
# level plans of the batched path on big shards now that the scan tightens by itself
mkdir -p gpurun_out
for cfg in "12500000 '' 1" "12500000 1 1" "12500000 4 1" "12500000 '' 0" "125000000 '' 1" "125000000 8 1" "125000000 16 1"; do
eval set -- $cfg
DPQ_PLAN_RATIOS=$2 DPQ_TIGHTEN=$3 timeout -k 10 600 python bench.py --codes $1 --data stream --steps 5 --warmup 1 --reps 3 --check 4 --no-cpu-baseline --sustain-seconds 0 --host-steps 0 > gpurun_out/bp.json 2>gpurun_out/bp.err || { tail -5 gpurun_out/bp.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/bp.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("codes $1 ratios '$2' tighten $3:", round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step scan", round(r["scan_ms_per_step"],3), "sel", round(r["select_ms_per_step"],3), "launches", r["launches_per_step"], "checks/q", round(r["filter_survivors_per_query"]), "cand/q", round(r["candidates_per_query"]), "parity", d["parity_checked_queries"], flush=True)
PY
done 2>&1 | tee gpurun_out/big_plans.txt
