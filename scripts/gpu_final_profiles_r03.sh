#!/bin/bash
# Final evidence of round 3: default bench line, kernel stats, PMC (default + the 125 M stream pass), SQ counters, the
# M = 16 and 12.5 M / 125 M lines.  Summaries land under gpurun_out/r03final/ and gpurun_out/r03_*/ (copy into profiles/).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03final
mkdir -p $O
step() { echo "$1" >> $O/progress.txt; }
step "bench default"; timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
step "pmc default"; bash scripts/collect_pmc.sh r03_default > $O/pmc_default.log 2>&1; echo "pmc default rc=$?"
step "sq"; bash scripts/collect_sq_pmc.sh r03_sq > $O/sq.log 2>&1; echo "sq rc=$?"
step "pmc 125M stream"; bash scripts/collect_pmc.sh r03_125M_stream --codes 125000000 --data stream --queries 1 --check 8 > $O/pmc_125M_stream.log 2>&1; echo "pmc 125M stream rc=$?"
for q in 2 4; do step "125M stream q=$q"; timeout -k 10 500 python bench.py --codes 125000000 --data stream --queries $q --steps 10 --warmup 2 --reps 5 --check 8 --no-cpu-baseline > $O/bench_125M_stream_q$q.json 2> $O/bench_125M_q$q.err; echo "125M q=$q rc=$?"; done
step "m16"; BENCH_ARGS="--m 16 --topk 1000" bash scripts/gpu_kstats.sh m16_top1000 > $O/kstats_m16.log 2>&1
timeout -k 10 600 python bench.py --m 16 --topk 1000 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_m16_top1000.json 2> $O/bench_m16.err; echo "m16 rc=$?"
step "12.5M"; timeout -k 10 600 python bench.py --codes 12500000 --data stream --steps 10 --warmup 2 --reps 5 --check 8 --no-cpu-baseline > $O/bench_12p5M.json 2> $O/bench_12p5M.err; echo "12.5M rc=$?"
step "125M batched"; timeout -k 10 700 python bench.py --codes 125000000 --data stream --steps 5 --warmup 1 --reps 3 --check 8 --no-cpu-baseline > $O/bench_125M.json 2> $O/bench_125M.err; echo "125M rc=$?"
for f in bench_default bench_m16_top1000 bench_12p5M bench_125M bench_125M_stream_q2 bench_125M_stream_q4; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$f', round(d['value']), round(d['ms_per_step'],4), 'parity', d['parity_checked_queries'], r['bound'], 'frac', round(r['frac'],3), 'scan', round(r['scan_ms_per_step'],4), 'sel', round(r['select_ms_per_step'],4), 'cand', round(r['candidates_per_query']), 'cpu', round(d['cpu_baseline']['value'],1) if 'cpu_baseline' in d else None)"; done
tail -6 $O/pmc_default.log; tail -6 $O/pmc_125M_stream.log; tail -30 $O/sq.log
