#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
mkdir -p gpurun_out
for cap in 3072 4096 6144 3072 4096 6144; do
DPQ_BOOT_CAP=$cap python bench.py --no-cpu-baseline --reps 5 > gpurun_out/bc.json 2>gpurun_out/bc.err || { tail -5 gpurun_out/bc.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/bc.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("cap $cap", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "boot+sel", round(r["select_ms_per_step"],4), "checks/q", round(r["filter_survivors_per_query"]), "cand/q", round(r["candidates_per_query"]), flush=True)
PY
done
