#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02g
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for ov in 1 0; do
  DPQ_ASYNC_OVERLAP=$ov timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02g/overlap$ov.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r02g/overlap$ov.json'));r=d['roofline'];print('overlap=$ov', round(d['value']), round(d['ms_per_step'],4), d['repetitions']['ms_per_step_min'], d['repetitions']['ms_per_step_max'], 'scan', round(r['scan_ms_per_step'],4), 'parity', d['parity_checked_queries'])"
done
