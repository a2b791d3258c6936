#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02i
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for fu in 1 0; do
  DPQ_FUSE_QUANTISE=$fu timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02i/fuse$fu.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r02i/fuse$fu.json'));r=d['roofline'];print('fuse=$fu', round(d['value']), round(d['ms_per_step'],4), 'scan', round(r['scan_ms_per_step'],4), 'sel+boot', round(r['select_ms_per_step'],4), 'cand', round(r['candidates_per_query']), 'parity', d['parity_checked_queries'])"
done
DPQ_ASYNC_OVERLAP=0 bash scripts/gpu_kstats.sh fused_serial | grep -v lut_build
