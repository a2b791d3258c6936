#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# usage: gpu_kstats.sh TAG [ENV=VAL ...] -- per-kernel average times of the default bench under rocprofv3 --kernel-trace --stats
set -o pipefail
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
# per-kernel figures want one kernel at a time on the GPU: pipelined batches on one lane
export DPQ_ASYNC_OVERLAP=${DPQ_ASYNC_OVERLAP:-0}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/kstats_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg --check 0 $BENCH_ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || echo "rocprof failed"
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/stats
echo "== $TAG $*"
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    n = r["Name"]
    if "dpq::" in n and "anonymous" not in n and "encode_pq" not in n and "decode_segments" not in n:
        print("  %-44s calls %5d avg %9.1f us" % (n.replace("void dpq::", "").replace("dpq::", "").split("(")[0][:44], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
python - <<PY
import json
d=json.load(open("$OUT/bench_under_rocprof.json")); r=d["roofline"]
print("  value %.0f q/s  ms/step %.4f  cand/q %.0f  checks/q %.0f" % (d["value"], d["ms_per_step"], r["candidates_per_query"], r["filter_survivors_per_query"]))
PY
