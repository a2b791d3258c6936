#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# top-1000: bootstrap capacity beyond 8192 (M = 8 and M = 16)
mkdir -p gpurun_out
for m in 8 16; do
for cap in 0 12288 16384 24576; do
    DPQ_BOOT_CAP=$cap python bench.py --no-cpu-baseline --reps 3 --m $m --topk 1000 > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; continue; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/sweep.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("m $m cap $cap top1000", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "select+boot", round(r["select_ms_per_step"],4), "checks/q", round(r["filter_survivors_per_query"]), "cand/q", round(r["candidates_per_query"]), "launches", r["launches_per_step"], flush=True)
PY
done
done 2>&1 | tee gpurun_out/boot_cap1000.txt
