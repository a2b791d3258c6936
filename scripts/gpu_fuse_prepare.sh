#!/bin/bash
# the default bench workload with the batch's decode + table build as one launch (DPQ_FUSE_PREPARE=1, the default) or two,
# same box back to back; then M = 16 / top-1000 both ways
cd "$GRAFT_REPO_ROOT" || exit 1
export DPQ_DEV=1
for g in 1 0 1 0 1; do
  DPQ_FUSE_PREPARE=$g timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-hbm-leg --no-cpu-baseline --sustain-seconds 1 --host-steps 0 --check 8 > gpurun_out/fp.json 2> gpurun_out/fp.err || { tail -3 gpurun_out/fp.err; continue; }
  python -c "
import json;d=json.loads(open('gpurun_out/fp.json').read().strip().splitlines()[-1]);print('fuse $g:', round(d['value']), 'q/s', round(d['ms_per_step'],4), 'ms/step; sustained', round(d['sustained']['value']), 'min/max', round(d['repetitions']['value_min']), round(d['repetitions']['value_max']))"
done 2>&1 | tee gpurun_out/fuse_prepare.txt
for g in 1 0; do
  DPQ_FUSE_PREPARE=$g timeout -k 10 400 python bench.py --m 16 --topk 1000 --steps 10 --warmup 3 --no-hbm-leg --no-cpu-baseline --sustain-seconds 0 --host-steps 0 --check 4 --index-dir /tmp/dpq_index_cache > gpurun_out/fp.json 2> gpurun_out/fp.err || { tail -3 gpurun_out/fp.err; continue; }
  python -c "
import json;d=json.loads(open('gpurun_out/fp.json').read().strip().splitlines()[-1]);print('m16 top-1000 fuse $g:', round(d['value']), 'q/s', round(d['ms_per_step'],4), 'ms/step')"
done 2>&1 | tee -a gpurun_out/fuse_prepare.txt
