#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02e
run() { tag=$1; m=$2; k=$3; shift 3; env "$@" timeout -k 10 300 python bench.py --m $m --topk $k --steps 10 --warmup 2 --reps 3 --check 2 --no-cpu-baseline > gpurun_out/r02e/plan_$tag.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r02e/plan_$tag.json'));r=d['roofline'];print('$tag', round(d['value']), round(d['ms_per_step'],3), 'scan', round(r['scan_ms_per_step'],3), 'sel', round(r['select_ms_per_step'],3), 'launches', r['launches_per_step'], 'cand', round(r['candidates_per_query']))"; }
run m16k1000_333 16 1000 DPQ_PLAN_RATIOS=3,3,3
run m16k1000_222 16 1000 DPQ_PLAN_RATIOS=2,2,2
run m16k1000_24 16 1000 DPQ_PLAN_RATIOS=2,4
run m8k1000_single 8 1000 DPQ_X=1
run m8k1000_33 8 1000 DPQ_PLAN_RATIOS=3,3
run m8k1000_333 8 1000 DPQ_PLAN_RATIOS=3,3,3
run m8k10_single 8 10 DPQ_X=1
run m8k10_r4 8 10 DPQ_PLAN_RATIOS=4
run m8k300_single 8 300 DPQ_X=1
run m8k300_r4 8 300 DPQ_PLAN_RATIOS=4
run m8k300_33 8 300 DPQ_PLAN_RATIOS=3,3
