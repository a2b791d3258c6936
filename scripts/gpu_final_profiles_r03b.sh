#!/bin/bash
# Refresh of the round-3 evidence that depends on the final stream-pass kernels: default bench line (with hbm_regime),
# 125 M stream pass (kernel stats + PMC), its SQ counters, two / four queries per call, and the full GPU test log.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03final
mkdir -p $O
step() { echo "$1" >> $O/progress.txt; }
step "pytest"; python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
step "bench default"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
step "pmc 125M stream"; bash scripts/collect_pmc.sh r03_125M_stream --codes 125000000 --data stream --queries 1 --check 8 > $O/pmc_125M_stream.log 2>&1; echo "pmc 125M stream rc=$?"
step "sq strand"; KERNEL=strand_kernel bash scripts/collect_sq_pmc.sh r03_sq_strand --codes 125000000 --data stream --queries 1 > $O/sq_strand.log 2>&1; echo "sq strand rc=$?"
for q in 2 4; do step "125M stream q=$q"; timeout -k 10 500 python bench.py --codes 125000000 --data stream --queries $q --steps 10 --warmup 2 --reps 5 --check 8 --no-cpu-baseline > $O/bench_125M_stream_q$q.json 2> $O/bench_125M_q$q.err; echo "125M q=$q rc=$?"; done
for f in bench_default bench_125M_stream_q2 bench_125M_stream_q4; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$f', round(d['value']), round(d['ms_per_step'],4), 'parity', d['parity_checked_queries'], r['bound'], r['kernel'], 'frac', round(r['frac'],3), 'hbm_regime', (d.get('hbm_regime') or {}).get('frac'))"; done
tail -6 $O/pmc_125M_stream.log; grep "SQ_LDS_IDX_ACTIVE\|SQ_LDS_BANK_CONFLICT\|SQ_BUSY_CU_CYCLES\|SQ_INSTS_VALU \|SQ_INSTS_LDS \|mean_ns" $O/sq_strand.log
