#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r02c
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --check 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || echo "rocprof failed"
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/stats
cat $OUT/kernel_stats.csv | cut -c1-200
for cap in 2048 4096 6144; do
  DPQ_BOOT_CAP=$cap timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --check 2 > $OUT/bench_cap$cap.json 2>/dev/null
  echo "cap=$cap: $(python -c "import json;d=json.load(open('$OUT/bench_cap$cap.json'));r=d['roofline'];print(d['value'], r['scan_ms_per_step'], r['select_ms_per_step'], r['candidates_per_query'], r['filter_survivors_per_query'])")"
done
for ratios in 8 4 16; do
  DPQ_PLAN_RATIOS=$ratios timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --check 2 > $OUT/bench_ratio$ratios.json 2>/dev/null
  echo "ratios=$ratios: $(python -c "import json;d=json.load(open('$OUT/bench_ratio$ratios.json'));r=d['roofline'];print(d['value'], r['scan_ms_per_step'], r['select_ms_per_step'], r['candidates_per_query'], r['filter_survivors_per_query'])")"
done
