#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# bank-aware relabelling of the scratch's codes (default) against code values as labels (DPQ_RELABEL=0)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -1 gpurun_out/pytest_gpu.log
for v in 1 0 1 0; do
  for k in 100 10; do
  DPQ_RELABEL=$v python bench.py --no-cpu-baseline --reps 5 --topk $k > gpurun_out/rl.json 2>gpurun_out/rl.err || { tail -5 gpurun_out/rl.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/rl.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("relabel=$v top$k", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "decode", round(r["decode_ms_per_step"],4), "frac", round(r["frac"],3), flush=True)
PY
  done
done
