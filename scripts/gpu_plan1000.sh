#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# top-1000: cascade plans behind the bootstrap (M = 8 at cap 8192 and 12288, M = 16 at cap 12288)
mkdir -p gpurun_out
for cfg in "8 0" "8 12288" "16 12288"; do
set -- $cfg
for plan in "3,3" "4" "8" "2,4" "4,4" "2,2,2"; do
    DPQ_BOOT_CAP=$2 DPQ_PLAN_RATIOS=$plan python bench.py --no-cpu-baseline --reps 3 --m $1 --topk 1000 > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; continue; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/sweep.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("m $1 cap $2 plan $plan top1000", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "select+boot", round(r["select_ms_per_step"],4), "checks/q", round(r["filter_survivors_per_query"]), "cand/q", round(r["candidates_per_query"]), "launches", r["launches_per_step"], flush=True)
PY
done
done 2>&1 | tee gpurun_out/plan1000.txt
