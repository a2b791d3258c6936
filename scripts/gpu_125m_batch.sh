#!/bin/bash
mkdir -p gpurun_out
for mb in 256 1024; do
echo "budget $mb MB: start" >> gpurun_out/b125_progress.txt
DPQ_BATCH_RAW_MB=$mb timeout -k 10 500 python bench.py --codes 125000000 --data stream --steps 5 --warmup 2 --reps 3 --check 2 --no-cpu-baseline > gpurun_out/b125_$mb.json 2>gpurun_out/b125_$mb.err || { tail -5 gpurun_out/b125_$mb.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/b125_$mb.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("budget $mb", round(d["value"]), round(d["ms_per_step"],3), "scan", round(r["scan_ms_per_step"],3), "decode", round(r["decode_ms_per_step"],3), "sel", round(r["select_ms_per_step"],3), d["config"]["decode"][:60], flush=True)
PY
done
