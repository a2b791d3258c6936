#!/bin/bash
# the default bench workload: wave priority of the kernels that run under a scan (in-tree: 3, variant prio0: 0) x lane gate
cd "$GRAFT_REPO_ROOT" || exit 1
export DPQ_DEV=1
for rep in 1 2; do
for cfg in ":0" ":1" "variants/lib_prio0.so:0" "variants/lib_prio0.so:1"; do
  lib=${cfg%%:*}; g=${cfg##*:}
  DPQ_LIB_PATH=${lib:+$PWD/$lib} DPQ_LANE_GATE=$g timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-hbm-leg --no-cpu-baseline --sustain-seconds 1 --host-steps 0 --check 8 > gpurun_out/lg.json 2> gpurun_out/lg.err || { tail -3 gpurun_out/lg.err; continue; }
  python -c "
import json;d=json.loads(open('gpurun_out/lg.json').read().strip().splitlines()[-1]);print('${lib:-in-tree (prio 3)} gate $g:', round(d['value']), 'q/s', round(d['ms_per_step'],4), 'ms/step; sustained', round(d['sustained']['value']), 'min/max', round(d['repetitions']['value_min']), round(d['repetitions']['value_max']))"
done; done 2>&1 | tee gpurun_out/prio_ab.txt
