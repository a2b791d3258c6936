#!/bin/bash
# select_kernel with its second launch bound (four 512-thread blocks per CU): the new parity test, stamps and lines over top_k
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/sellb; mkdir -p $O
step() { echo "$(date +%T) $1" | tee -a $O/progress.txt; }
step "new test + subset"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_hand_derived.py -q -x -m gpu -k "histogram_selection or large_topk or tie_explosion or overflow or m16_sift1m or full_size_sift1m or one_query_per_pass or hand_derived or golden" > $O/pytest_subset.log 2>&1 || { tail -30 $O/pytest_subset.log; exit 1; }
tail -2 $O/pytest_subset.log
step "stamps"
CFGS="8:1000 8:2048 16:1000" bash scripts/gpu_select_stamps_large_k.sh > $O/stamps_wrap.log 2>&1; grep -E "===|select blocks|final select" gpurun_out/selk/stamps.txt
export DPQ_DEV=1
for cfg in "8 100" "8 512" "8 1000" "8 2048" "16 1000"; do
  set -- $cfg
  for rep in 1 2; do
    timeout -k 10 400 python bench.py --index-dir /tmp/dpq_index_cache --m $1 --topk $2 --no-cpu-baseline --reps 8 --no-hbm-leg --sustain-seconds 0 --host-steps 0 --no-replicas > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
    python - <<PY | tee -a $O/ab.txt
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("M=$1 top-$2:", round(d["value"]), "q/s", round(d["ms_per_step"],4), "ms/step; scan", round(r["scan_ms_per_step"],4), "select", round(r["select_ms_per_step"],4), flush=True)
PY
  done
done
step "one-query latency, 1 M codes"
timeout -k 10 300 python scripts/dev_latency_big.py 1000000 2>&1 | grep -v amdgpu.ids | tee -a $O/latency.txt
step done
