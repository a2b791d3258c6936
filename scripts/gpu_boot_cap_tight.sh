#!/bin/bash
export DPQ_DEV=1
# bootstrap size against the in-scan tightening: how many nodes the bootstrap should evaluate now that the scan tightens by itself
mkdir -p gpurun_out
for cfg in "0 1" "2048 1" "2560 1" "4096 1" "0 0" "2048 0"; do
set -- $cfg
DPQ_BOOT_CAP=$1 DPQ_TIGHTEN=$2 python bench.py --no-cpu-baseline --reps 6 > gpurun_out/bc.json 2>gpurun_out/bc.err || { tail -5 gpurun_out/bc.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/bc.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("boot_cap $1 tighten $2:", round(d["value"]), "q/s", round(d["ms_per_step"],4), "ms/step scan", round(r["scan_ms_per_step"],4), "select+boot", round(r["select_ms_per_step"],4), "checks/q", round(r["filter_survivors_per_query"]), "cand/q", round(r["candidates_per_query"]), flush=True)
PY
done 2>&1 | tee gpurun_out/boot_cap_tight.txt
