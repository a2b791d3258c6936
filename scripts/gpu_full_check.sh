#!/bin/bash
# full GPU suite, fuzz (small + bootstrap-sized shards), the default bench line (files under gpurun_out/full/)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/full; mkdir -p $O
( timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log )
( timeout -k 10 200 python scripts/fuzz_parity.py ${FUZZ_S:-100} 21 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log )
( DPQ_FUZZ_BIG=1 timeout -k 10 300 python scripts/fuzz_parity.py ${FUZZ_S:-100} 22 > $O/fuzz_big.log 2>&1; tail -1 $O/fuzz_big.log; grep -c "MISMATCH\|ERROR" $O/fuzz_big.log )
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(d['ms_per_step'],4), 'parity', d['parity_checked_queries'], r['bound'], 'frac', round(r['frac'],3), 'scan', round(r['scan_ms_per_step'],4),
      'checks', round(r['filter_survivors_per_query']), 'cand', round(r['candidates_per_query']), 'sustained', round(d['sustained']['value']),
      'h2h', round(d['host_to_host']['value']), 'cpu', round(d['cpu_baseline']['value'],1))
print('hbm_regime', json.dumps(d.get('hbm_regime')))
PY
