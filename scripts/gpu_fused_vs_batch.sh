#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not bench and not 125m and not rccl" > gpurun_out/fb_pytest.log 2>&1 || { tail -30 gpurun_out/fb_pytest.log; exit 1; }
tail -1 gpurun_out/fb_pytest.log
for cfg in "--batch-decode -1" "--batch-decode 0" "--batch-decode -1 --topk 10" "--batch-decode 0 --topk 10" "--batch-decode -1 --m 16 --topk 1000" "--batch-decode 0 --m 16 --topk 1000"; do
python bench.py --no-cpu-baseline --reps 5 $cfg > gpurun_out/fb.json 2>gpurun_out/fb.err || { tail -5 gpurun_out/fb.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/fb.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$cfg", round(d["value"]), round(d["ms_per_step"],4), "scan", round(r["scan_ms_per_step"],4), "frac", round(r["frac"],3), flush=True)
PY
done
