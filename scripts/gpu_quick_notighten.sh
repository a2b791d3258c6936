#!/bin/bash
# the four quick bench lines without the in-scan tightening (A/B for scripts/gpu_quick.sh)
export DPQ_DEV=1 DPQ_TIGHTEN=0
for cfg in "--m 8 --topk 100" "--m 16 --topk 1000" "--m 8 --topk 10" "--m 8 --topk 1000"; do
python bench.py --no-cpu-baseline --reps 5 $cfg > gpurun_out/quick.json 2>gpurun_out/quick.err || { tail -5 gpurun_out/quick.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/quick.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("notighten $cfg", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "select+boot", round(r["select_ms_per_step"],4), "frac", round(r["frac"],3), "checks/q", round(r["filter_survivors_per_query"]), flush=True)
PY
done
