#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# A/B of the bootstrap's centroid visiting order: neighbour lists (default) against the exact 256-key sorts.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/ab_pytest.log 2>&1 || { tail -30 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
for v in 0 1 0 1; do
  DPQ_BOOT_FULLSORT=$v python bench.py --no-cpu-baseline --reps 6 > gpurun_out/ab_full$v.json 2>gpurun_out/ab_full$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_full$v.json").read().strip().splitlines()[-1])
print("fullsort=$v", d["value"], d["ms_per_step"], d.get("kernels_ms_per_step"))
PY
done
DPQ_BOOT_FULLSORT=0 python bench.py --no-cpu-baseline --reps 4 --m 16 --topk 1000 > gpurun_out/ab_m16_0.json 2>/dev/null && DPQ_BOOT_FULLSORT=1 python bench.py --no-cpu-baseline --reps 4 --m 16 --topk 1000 > gpurun_out/ab_m16_1.json 2>/dev/null
python - <<PY
import json
for v in (0,1):
    d=json.loads(open(f"gpurun_out/ab_m16_{v}.json").read().strip().splitlines()[-1])
    print("m16 fullsort", v, d["value"], d["ms_per_step"])
PY
