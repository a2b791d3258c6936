#!/bin/bash
export DPQ_DEV=1
# bootstrap size for a large top_k now that one level + in-scan tightening is the plan there
mkdir -p gpurun_out
for cfg in "--m 8 --topk 1000" "--m 16 --topk 1000" "--m 8 --topk 512" "--m 8 --topk 2048"; do
for cap in 4096 6144 8192 12288; do
DPQ_BOOT_CAP=$cap python bench.py --no-cpu-baseline --reps 4 $cfg > gpurun_out/bc.json 2>gpurun_out/bc.err || { tail -5 gpurun_out/bc.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/bc.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$cfg boot_cap $cap:", round(d["value"]), "q/s", round(d["ms_per_step"],4), "ms/step scan", round(r["scan_ms_per_step"],4), "select+boot", round(r["select_ms_per_step"],4), "checks/q", round(r["filter_survivors_per_query"]), "cand/q", round(r["candidates_per_query"]), flush=True)
PY
done
done 2>&1 | tee gpurun_out/boot_cap_large_k.txt
