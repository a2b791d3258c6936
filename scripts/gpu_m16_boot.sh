#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# M = 16: bootstrap capacity / target against the step time (top-100 and top-1000)
mkdir -p gpurun_out
for cfg in "0 0" "4096 0" "6144 0" "8192 0" "3072 2048" ; do
  set -- $cfg
  for k in 100 1000; do
    DPQ_BOOT_CAP=$1 DPQ_BOOT_TARGET=$2 python bench.py --no-cpu-baseline --reps 3 --m 16 --topk $k > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; continue; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/sweep.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("cap $1 target $2 top$k", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "select+boot", round(r["select_ms_per_step"],4), "checks/q", round(r["filter_survivors_per_query"]), "cand/q", round(r["candidates_per_query"]), flush=True)
PY
  done
done 2>&1 | tee gpurun_out/m16_boot.txt
