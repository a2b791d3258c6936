#!/bin/bash
# A/B of libraries under variants/ on the default workload (and top-10: the filter almost alone)
mkdir -p gpurun_out
for lib in variants/lib_*.so; do
  for k in 100 10; do
    DPQ_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --reps 5 --topk $k --check 16 > gpurun_out/var.json 2>gpurun_out/var.err || { tail -5 gpurun_out/var.err; continue; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/var.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$lib top$k", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "boot+sel", round(r["select_ms_per_step"],4), "checks/q", round(r["filter_survivors_per_query"]), flush=True)
PY
  done
done 2>&1 | tee gpurun_out/variants.txt
