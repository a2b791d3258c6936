#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02e
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --m 16 --topk 1000 --steps 10 --warmup 2 --reps 3 --check 4 --no-cpu-baseline $EXTRA > gpurun_out/r02e/sweep_$tag.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r02e/sweep_$tag.json'));r=d['roofline'];print('$tag', round(d['value']), round(d['ms_per_step'],3), 'scan', round(r['scan_ms_per_step'],3), 'sel', round(r['select_ms_per_step'],3), 'launches', r['launches_per_step'], 'cand', round(r['candidates_per_query']))"; }
run single DPQ_X=1
run r8 DPQ_PLAN_RATIOS=8
run r4_4 DPQ_PLAN_RATIOS=4,4
run r16 DPQ_PLAN_RATIOS=16
run r3_3 DPQ_PLAN_RATIOS=3,3
EXTRA="--bootstrap -1" run classic DPQ_X=1
