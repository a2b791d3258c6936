#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) of the default bench.py workload, summarised into
# gpurun_out/<tag>_*.  PMC passes use only --kernel-trace, as the pool requires.
#   gpurun -- 'bash scripts/collect_pmc.sh r01_v3'
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.err || exit 2
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err || exit 3
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err || exit 4
python scripts/summarise_pmc.py $OUT $TAG
# keep the summaries only: the raw traces are tens of MB and gpurun_out/ must stay under 64 MiB
cp $OUT/pmc_fetch/*/*_counter_collection.csv $OUT/${TAG}_fetch_counter_collection.csv 2>/dev/null
cp $OUT/pmc_write/*/*_counter_collection.csv $OUT/${TAG}_write_counter_collection.csv 2>/dev/null
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
