#!/bin/bash
# experiment: decode once per batch into plain codes (DPQ_BATCH_DECODE=1) against the fused decode
mkdir -p gpurun_out
DPQ_BATCH_DECODE=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not bench and not 125m and not rccl" > gpurun_out/bd_pytest.log 2>&1 || { tail -30 gpurun_out/bd_pytest.log; exit 1; }
tail -2 gpurun_out/bd_pytest.log
for v in 0 1 0 1; do
  for cfg in "--topk 100" ; do
  DPQ_BATCH_DECODE=$v python bench.py --no-cpu-baseline --reps 5 $cfg > gpurun_out/bd.json 2>gpurun_out/bd.err || { tail -5 gpurun_out/bd.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bd.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("batch_decode=$v $cfg", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "boot+sel", round(r["select_ms_per_step"],4), flush=True)
PY
  done
done
for v in 0 1; do
  DPQ_BATCH_DECODE=$v python bench.py --no-cpu-baseline --reps 3 --m 16 --topk 1000 > gpurun_out/bd.json 2>gpurun_out/bd.err || { tail -5 gpurun_out/bd.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bd.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("batch_decode=$v m16 top1000", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), flush=True)
PY
done
