#!/bin/bash
# round 4: the select kernel's last level as a bucket sort (SelectArgs.fast_final) against the radix select + rank / bitonic sort,
# over top_k; parity subset first
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/selb; mkdir -p $O
step() { echo "$(date +%T) $1" | tee -a $O/progress.txt; }
step "parity subset"
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_hand_derived.py -q -x -m gpu -k "threshold_bootstrap or in_scan_tightening or m16_parity or m16_sift1m or full_size_sift1m or large_topk or tie_explosion or duplicate_heavy or parity_with_oracle or golden or overflow or shard_smaller or hand_derived or sharded_index or one_query" > $O/pytest_subset.log 2>&1 || { tail -30 $O/pytest_subset.log; exit 1; }
tail -2 $O/pytest_subset.log
export DPQ_DEV=1
step "stamps"
timeout -k 10 300 python scripts/dev_boot_stamps.py 1:::0 1:::1 > $O/stamps.txt 2>&1 || { tail -20 $O/stamps.txt; exit 1; }
grep -v "amdgpu.ids" $O/stamps.txt
line() {  # tag, select_fast, bench args
  local tag=$1 sf=$2; shift 2
  DPQ_SELECT_FAST=$sf timeout -k 10 400 python bench.py --index-dir /tmp/dpq_index_cache --no-cpu-baseline --reps ${REPS:-10} --no-hbm-leg --sustain-seconds 0 --host-steps 0 --no-replicas "$@" > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; return 1; }
  python - <<PY | tee -a $O/ab.txt
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
r=d["roofline"]; rp=d["repetitions"]
print("$tag fast=$sf:", round(d["value"]), "q/s", round(d["ms_per_step"],4), "ms/step (min %.4f); scan" % rp["ms_per_step_min"], round(r.get("scan_ms_per_step",0),4), "select", round(r.get("select_ms_per_step",0),4), "checks/q", round(r.get("filter_survivors_per_query",0)), "cand/q", round(r.get("candidates_per_query",0)), flush=True)
PY
}
for k in 100 300 512 1000 2048; do
  for sf in 0 1 0 1; do step "top-$k fast=$sf"; line "M=8 top-$k" $sf --topk $k || exit 1; done
done
for sf in 0 1 0 1; do step "m16 top-1000 fast=$sf"; line "M=16 top-1000" $sf --m 16 --topk 1000 || exit 1; done
for sf in 0 1; do step "m16 top-100 fast=$sf"; line "M=16 top-100" $sf --m 16 --topk 100 || exit 1; done
step done
