#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
export DPQ_ASYNC_OVERLAP=0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/dbg_rp; rm -rf $OUT; mkdir -p $OUT
( while true; do sleep 30; date >> $OUT/heartbeat.txt; done ) &
HB=$!
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --reps 2 --no-hbm-leg --sustain-seconds 0 --host-steps 0 > $OUT/line.json 2> $OUT/stats.err; echo "stats rc=$?"
tail -5 $OUT/stats.err
kill $HB
