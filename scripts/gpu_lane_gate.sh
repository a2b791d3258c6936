#!/bin/bash
# the default bench workload by DPQ_LANE_GATE (0 never, 1 second batch of a burst, 2 every batch), same box back to back
cd "$GRAFT_REPO_ROOT" || exit 1
export DPQ_DEV=1
for g in 1 0 2 1 0; do
  DPQ_LANE_GATE=$g timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-hbm-leg --no-cpu-baseline --sustain-seconds 1 --host-steps 0 --check 8 > gpurun_out/lg.json 2> gpurun_out/lg.err || { tail -3 gpurun_out/lg.err; continue; }
  python -c "
import json;d=json.loads(open('gpurun_out/lg.json').read().strip().splitlines()[-1]);print('gate $g:', round(d['value']), 'q/s', round(d['ms_per_step'],4), 'ms/step; sustained', round(d['sustained']['value']), 'min/max', round(d['repetitions']['value_min']), round(d['repetitions']['value_max']))"
done 2>&1 | tee gpurun_out/lane_gate.txt
