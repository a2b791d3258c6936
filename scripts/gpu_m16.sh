#!/bin/bash
# M = 16 after the move to 8-bit filter entries: parity tests of the M = 16 cases, then the configs[2] bench line.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "m16 or M16 or other_codebook or hand" > gpurun_out/m16_pytest.log 2>&1 || { tail -40 gpurun_out/m16_pytest.log; exit 1; }
tail -2 gpurun_out/m16_pytest.log
python bench.py --no-cpu-baseline --reps 5 --m 16 --topk 1000 > gpurun_out/m16_line.json 2>gpurun_out/m16_line.err || { tail -20 gpurun_out/m16_line.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/m16_line.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("m16 top1000", d["value"], d["ms_per_step"], "scan", r["scan_ms_per_step"], "select", r["select_ms_per_step"], "checks/q", r["filter_survivors_per_query"], "cand/q", r["candidates_per_query"], "launches", r["launches_per_step"])
PY
python bench.py --no-cpu-baseline --reps 5 --m 16 --topk 100 > gpurun_out/m16_line100.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/m16_line100.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("m16 top100", d["value"], d["ms_per_step"], "scan", r["scan_ms_per_step"], "select", r["select_ms_per_step"], "checks/q", r["filter_survivors_per_query"], "cand/q", r["candidates_per_query"], "launches", r["launches_per_step"])
PY
