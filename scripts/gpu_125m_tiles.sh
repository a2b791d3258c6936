#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
mkdir -p gpurun_out
for tn in 0 33554432 8388608; do
DPQ_BATCH_TILE_NODES=$tn timeout -k 10 500 python bench.py --codes 125000000 --data stream --steps 5 --warmup 2 --reps 3 --check 2 --no-cpu-baseline > gpurun_out/b125_t.json 2>gpurun_out/b125_t.err || { tail -5 gpurun_out/b125_t.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/b125_t.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("tile nodes $tn", round(d["value"]), round(d["ms_per_step"],3), "scan", round(r["scan_ms_per_step"],3), "decode", round(r["decode_ms_per_step"],3), "launches", r["launches_per_step"], d["config"]["decode"][:70], flush=True)
PY
done
