#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a
# bench.py workload, summarised into gpurun_out/<tag>/.  PMC passes use only --kernel-trace, as the pool requires.
#   gpurun -- 'bash scripts/collect_pmc.sh r02_default'
#   gpurun -- 'bash scripts/collect_pmc.sh r02_125M --codes 125000000 --data stream'
set -o pipefail
TAG=${1:-r02}; shift
# per-kernel figures want one kernel at a time on the GPU: pipelined batches on one lane
export DPQ_ASYNC_OVERLAP=${DPQ_ASYNC_OVERLAP:-0}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
# the raw traces are tens of MB and gpurun_out/ must stay under 64 MiB (or nothing of the call is copied back): drop them however this ends
trap 'rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write' EXIT
COMMON="--no-cpu-baseline --reps 2 $*"
# a pipeline-built index is built ONCE, outside the profiled processes (bench.py --index-dir): under per-dispatch counter
# collection the M = 16 builder's millions of dispatches never finished
case " $* " in *" --data stream "*) ;; *)
  COMMON="$COMMON --index-dir /tmp/dpq_index_cache"
  echo "pass 0: index build (unprofiled)" >> $OUT/progress.txt
  timeout -k 10 600 python bench.py --build-only --index-dir /tmp/dpq_index_cache "$@" > $OUT/build.json 2> $OUT/build.err || exit 1;;
esac
echo "pass 1: kernel stats" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 3 --no-hbm-leg --sustain-seconds 0 --host-steps 0 $COMMON > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.err || exit 2
echo "pass 2: FETCH_SIZE" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 5 --warmup 1 --check 0 --sustain-seconds 0 --host-steps 0 --no-hbm-leg $COMMON > /dev/null 2> $OUT/fetch.err || exit 3
echo "pass 3: WRITE_SIZE" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 5 --warmup 1 --check 0 --sustain-seconds 0 --host-steps 0 --no-hbm-leg $COMMON > /dev/null 2> $OUT/write.err || exit 4
python scripts/summarise_pmc.py $OUT $TAG
# keep the summaries only: the raw traces are tens of MB and gpurun_out/ must stay under 64 MiB
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
