#!/bin/bash
# Round-3 closing evidence after the large-top_k plan change (the full GPU suite and a fuzz run went before it:
# 118 passed, 19 522 fuzz cases, 0 bad): the M = 16 top-1000 line with kernel stats + PMC passes, and the M = 8
# top-1000 / top-2048 lines.  (The default workload's kernels are unchanged: its profiles stay.)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03final
rm -rf gpurun_out/* && mkdir -p $O
step() { echo "$1" >> $O/progress.txt; }
step "m16 line"; timeout -k 10 600 python bench.py --m 16 --topk 1000 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_m16_top1000.json 2> $O/bench_m16.err; echo "m16 rc=$?"
for k in 1000 2048; do step "top-$k"; timeout -k 10 600 python bench.py --topk $k --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_top$k.json 2> $O/bench_top$k.err; echo "top-$k rc=$?"; done
step "pmc m16"; bash scripts/collect_pmc.sh r03_m16_top1000 --m 16 --topk 1000 > $O/pmc_m16.log 2>&1; echo "pmc m16 rc=$?"
tail -3 gpurun_out/r03_m16_top1000/fetch.err | cut -c1-300
for f in bench_m16_top1000 bench_top1000 bench_top2048; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$f', round(d['value']), round(d['ms_per_step'],4), 'parity', d['parity_checked_queries'], r['bound'], 'frac', round(r['frac'],3), 'scan', round(r['scan_ms_per_step'],4), 'sel', round(r['select_ms_per_step'],4), 'checks', round(r['filter_survivors_per_query']), 'cand', round(r['candidates_per_query']))"; done
tail -6 $O/pmc_m16.log
du -sh gpurun_out
