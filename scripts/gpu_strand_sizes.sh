#!/bin/bash
export DPQ_DEV=1
# where the strand image starts to pay: one and four queries per call on shards of growing size, both stream passes
mkdir -p gpurun_out
for n in 12500000 32000000 125000000; do for q in 1 4; do for st in 1 2; do
DPQ_STRANDS=$st timeout -k 10 400 python bench.py --codes $n --data stream --queries $q --steps 20 --warmup 3 --reps 3 --check 1 --no-cpu-baseline --sustain-seconds 0 --host-steps 0 > gpurun_out/sz.json 2>gpurun_out/sz.err || { tail -3 gpurun_out/sz.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/sz.json").read().strip().splitlines()[-1])
print("codes $n queries $q strands $st: %.1f us per call, launches %s" % (d["ms_per_step"]*1e3, d["roofline"]["launches_per_step"]), flush=True)
PY
done; done; done 2>&1 | tee gpurun_out/strand_sizes.txt
