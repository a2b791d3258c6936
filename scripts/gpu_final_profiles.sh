#!/bin/bash
# Final evidence of the round: default bench line, kernel stats, PMC (default + 125 M), SQ counters, M=16 and 12.5 M lines.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02final
echo "start bench default" > gpurun_out/r02final/progress.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02final/bench_default.json 2> gpurun_out/r02final/bench_default.err; echo "default rc=$?"
bash scripts/collect_pmc.sh r02_default > gpurun_out/r02final/pmc_default.log 2>&1; echo "pmc default rc=$?"
echo "pmc 125M start" >> gpurun_out/r02final/progress.txt; bash scripts/collect_pmc.sh r02_125M --codes 125000000 --data stream --check 2 > gpurun_out/r02final/pmc_125M.log 2>&1; echo "pmc 125M rc=$?"
echo "sq start" >> gpurun_out/r02final/progress.txt; bash scripts/collect_sq_pmc.sh r02_sq > gpurun_out/r02final/sq.log 2>&1; echo "sq rc=$?"
BENCH_ARGS="--m 16 --topk 1000" bash scripts/gpu_kstats.sh m16_top1000 > gpurun_out/r02final/kstats_m16.log 2>&1
timeout -k 10 600 python bench.py --m 16 --topk 1000 --steps 20 --warmup 3 > gpurun_out/r02final/bench_m16_top1000.json 2> gpurun_out/r02final/bench_m16.err; echo "m16 rc=$?"
echo "12.5M start" >> gpurun_out/r02final/progress.txt; timeout -k 10 600 python bench.py --codes 12500000 --data stream --steps 10 --warmup 2 --reps 5 --check 8 > gpurun_out/r02final/bench_12p5M.json 2> gpurun_out/r02final/bench_12p5M.err; echo "12.5M rc=$?"
for f in bench_default bench_m16_top1000 bench_12p5M; do python -c "
import json;d=json.load(open('gpurun_out/r02final/$f.json'));r=d['roofline'];print('$f', round(d['value']), round(d['ms_per_step'],4), 'parity', d['parity_checked_queries'], 'frac', round(r['frac'],3), 'scan', round(r['scan_ms_per_step'],4), 'sel', round(r['select_ms_per_step'],4), 'cand', round(r['candidates_per_query']), 'cpu', round(d['cpu_baseline']['value'],1) if 'cpu_baseline' in d else None)"; done
tail -8 gpurun_out/r02final/pmc_default.log; tail -8 gpurun_out/r02final/pmc_125M.log
