#!/bin/bash
# round 4, bootstrap_kernel<M, V>: parity of the default variant, phase stamps per variant, A/B of the pipelined step
# CONFIGS="variant:target:cap ..." (0 = default)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/bootv; mkdir -p $O
step() { echo "$(date +%T) $1" | tee -a $O/progress.txt; }
step "parity subset"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "threshold_bootstrap or in_scan_tightening or m16_parity or full_size_sift1m or large_topk or tie_explosion or duplicate_heavy or parity_with_oracle or golden" > $O/pytest_subset.log 2>&1 || { tail -30 $O/pytest_subset.log; exit 1; }
tail -2 $O/pytest_subset.log
step "stamps"
timeout -k 10 300 python scripts/dev_boot_stamps.py ${STAMPS:-0:::0 1:::0 1:::1 1:2048::1} > $O/stamps.txt 2>&1 || { tail -20 $O/stamps.txt; exit 1; }
grep -v "amdgpu.ids" $O/stamps.txt
step "index"
export DPQ_DEV=1
timeout -k 10 300 python bench.py --index-dir /tmp/dpq_index_cache --build-only > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
for cfg in ${CONFIGS:-1:::0 1:::1 1:::0 1:::1 1:::0 1:::1 0:::0 1:2048::1 0:::0 1:2048::1}; do
  IFS=: read v tg cp sf <<< "$cfg"
  DPQ_BOOT_VARIANT=$v DPQ_BOOT_TARGET=${tg:-0} DPQ_BOOT_CAP=${cp:-0} DPQ_SELECT_FAST=${sf:-1} timeout -k 10 200 python bench.py --index-dir /tmp/dpq_index_cache --no-cpu-baseline --reps ${REPS:-16} --no-hbm-leg --sustain-seconds 0 --host-steps 0 --no-replicas > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python - <<PY | tee -a $O/ab.txt
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
r=d["roofline"]; rp=d["repetitions"]
print("cfg $cfg:", round(d["value"]), "q/s", round(d["ms_per_step"],4), "ms/step (min %.4f max %.4f); scan" % (rp["ms_per_step_min"], rp["ms_per_step_max"]), round(r.get("scan_ms_per_step",0),4), "select+boot", round(r.get("select_ms_per_step",0),4), "checks/q", round(r.get("filter_survivors_per_query",0)), "cand/q", round(r.get("candidates_per_query",0)), flush=True)
PY
done
if [ -n "$SQ" ]; then
  step "sq counters of bootstrap and select"
  DPQ_BOOT_VARIANT=1 KERNEL=bootstrap_kernel,select_kernel,scan_kernel bash scripts/collect_sq_pmc.sh bootv_sq --index-dir /tmp/dpq_index_cache > $O/sq.log 2>&1; tail -150 $O/sq.log
fi
step done
