#!/bin/bash
# the secondary figures DESIGN.md section 7 quotes: top_k variants, small shards, small-batch latency
mkdir -p gpurun_out
{
for cfg in "--topk 10" "--topk 300" "--topk 1000" "--codes 125000" "--codes 250000" "--codes 500000"; do
python bench.py --no-cpu-baseline --reps 5 --check 8 $cfg > gpurun_out/dn.json 2>gpurun_out/dn.err || { tail -5 gpurun_out/dn.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/dn.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$cfg", round(d["value"]), "q/s", round(d["ms_per_step"],4), "ms/step scan", round(r["scan_ms_per_step"],4), "launches", r["launches_per_step"], d["config"]["decode"][:40], flush=True)
PY
done
python scripts/dev_latency.py 2>&1 | grep "nq="
} | tee gpurun_out/doc_numbers.txt
