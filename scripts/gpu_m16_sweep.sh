#!/bin/bash
# M = 16 filter geometry sweep: libraries built with -DDPQ_PRE16/-DDPQ_SAT16/-DDPQ_QT16 under variants/
mkdir -p gpurun_out
for lib in variants/lib_*.so; do
  for k in 1000 100; do
    DPQ_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --reps 3 --m 16 --topk $k > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; continue; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/sweep.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$lib top$k", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "checks/q", round(r["filter_survivors_per_query"]), "cand/q", round(r["candidates_per_query"]), flush=True)
PY
  done
done 2>&1 | tee gpurun_out/m16_sweep.txt
