#!/bin/bash
# full GPU suite, a short fuzz run, the default bench line (files under gpurun_out/final/)
set -o pipefail
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout -k 10 200 python scripts/fuzz_parity.py 60 11 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);r=d['roofline'];print(round(d['value']), round(d['ms_per_step'],4), 'parity', d['parity_checked_queries'], r['bound'], 'frac', round(r['frac'],3), 'scan', round(r['scan_ms_per_step'],4), 'checks', round(r['filter_survivors_per_query']), 'cand', round(r['candidates_per_query']), 'sustained', round(d['sustained']['value']), 'h2h', round(d['host_to_host']['value']), 'hbm_regime', (d.get('hbm_regime') or {}).get('frac'), 'cpu', round(d['cpu_baseline']['value'],1))"
