#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02h
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cap in 6144 4096 3072; do
  DPQ_BOOT_CAP=$cap timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02h/cap$cap.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r02h/cap$cap.json'));r=d['roofline'];print('cap=$cap', round(d['value']), round(d['ms_per_step'],4), 'scan', round(r['scan_ms_per_step'],4), 'sel+boot', round(r['select_ms_per_step'],4), 'cand', round(r['candidates_per_query']), 'checks', round(r['filter_survivors_per_query']), 'parity', d['parity_checked_queries'])"
done
bash scripts/gpu_kstats.sh boot32 | grep -v lut_build
