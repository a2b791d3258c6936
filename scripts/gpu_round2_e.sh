#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02e
BENCH_ARGS="--m 16 --topk 1000" bash scripts/gpu_kstats.sh m16_top1000
cp gpurun_out/kstats_m16_top1000/kernel_stats.csv gpurun_out/r02e/m16_top1000_kernel_stats.csv
timeout -k 10 600 python bench.py --m 16 --topk 1000 --steps 20 --warmup 3 > gpurun_out/r02e/bench_m16_top1000.json 2> gpurun_out/r02e/bench_m16.err; echo "m16 rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r02e/bench_m16_top1000.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], d['parity_checked_queries'], r['frac'], r['scan_ms_per_step'], r['select_ms_per_step'], r['candidates_per_query'], d['cpu_baseline']['value'])"
timeout -k 10 600 python bench.py --codes 12500000 --data stream --steps 10 --warmup 2 --reps 5 --check 8 > gpurun_out/r02e/bench_12p5M.json 2> gpurun_out/r02e/bench_12p5M.err; echo "12.5M rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r02e/bench_12p5M.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], d['parity_checked_queries'], r['frac'], r['scan_ms_per_step'], r['launches_per_step'], r['candidates_per_query'], d['cpu_baseline']['value'])"
