#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02f
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 3 --reps 5 --check 8 --no-cpu-baseline "$@" > gpurun_out/r02f/$tag.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r02f/$tag.json'));r=d['roofline'];print('$tag', round(d['value']), round(d['ms_per_step'],4), 'scan', round(r['scan_ms_per_step'],4), 'sel', round(r['select_ms_per_step'],4), 'launches', r['launches_per_step'], 'cand', round(r['candidates_per_query']), d['config']['threshold_bootstrap'])"; }
run n125k_auto --codes 125000
run n125k_off --codes 125000 --bootstrap -1
run n250k_auto --codes 250000
run n250k_off --codes 250000 --bootstrap -1
run n500k_auto --codes 500000
