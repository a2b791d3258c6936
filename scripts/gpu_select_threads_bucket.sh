#!/bin/bash
# the select block size again, now that the last level is a bucket sort (DPQ_SELECT_THREADS: 0 = by top_k)
cd "$GRAFT_REPO_ROOT" || exit 1
export DPQ_DEV=1
O=gpurun_out/selt; mkdir -p $O
for cfg in "8 1000" "8 2048" "16 1000" "8 512" "8 100"; do
  set -- $cfg
  for t in 256 512 256 512; do
    DPQ_SELECT_THREADS=$t timeout -k 10 400 python bench.py --index-dir /tmp/dpq_index_cache --m $1 --topk $2 --no-cpu-baseline --reps 8 --no-hbm-leg --sustain-seconds 0 --host-steps 0 --no-replicas > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
    python - <<PY | tee -a $O/ab.txt
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("M=$1 top-$2 select threads $t:", round(d["value"]), "q/s", round(d["ms_per_step"],4), "ms/step; scan", round(r["scan_ms_per_step"],4), "select", round(r["select_ms_per_step"],4), flush=True)
PY
  done
done
