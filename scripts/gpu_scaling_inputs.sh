#!/bin/bash
# inputs of the predicted 1 -> 8 GPU curve: what one rank of an index-sharded run spends per step on a 1/8,
# 1/4, 1/2 shard of the 1 M-code bench index (pipelined steps alone, and stream-ordered steps + pack + emulated 8-rank
# gather + merge), and the whole index
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/scaling; : > gpurun_out/scaling/inputs.txt
for n in 125000 250000 500000 1000000; do
  timeout -k 10 300 python bench.py --codes $n --steps 20 --warmup 3 --reps 5 --check 8 --no-cpu-baseline --no-hbm-leg --sustain-seconds 0 --host-steps 0 > gpurun_out/scaling/n$n.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/scaling/n$n.json').read().strip().splitlines()[-1]);r=d['roofline'];print('shard of $n codes, pipelined steps alone:', round(d['ms_per_step'],4), 'ms/step; scan', round(r['scan_ms_per_step'],4), 'select', round(r['select_ms_per_step'],4), 'lut', round(r['lut_ms_per_step'],4))" | tee -a gpurun_out/scaling/inputs.txt
  if [ $n -lt 1000000 ]; then N=$n timeout -k 10 300 python scripts/dev_sharded_step.py 2>/dev/null | tail -2 | tee -a gpurun_out/scaling/inputs.txt; fi
done
exit 0
