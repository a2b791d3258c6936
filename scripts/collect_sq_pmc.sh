#!/bin/bash
export DPQ_DEV=1   # developer switches of the library are read only with this set
# Runs ON THE GPU BOX (via gpurun): SQ / LDS counters of the scan kernel on a bench.py workload,
# one rocprofv3 --pmc pass per counter group (only --kernel-trace beside it, as the pool requires).
#   gpurun -- 'bash scripts/collect_sq_pmc.sh r02_sq [bench args]'
set -o pipefail
TAG=${1:-sq}; shift
# per-kernel figures want one kernel at a time on the GPU: pipelined batches on one lane
export DPQ_ASYNC_OVERLAP=${DPQ_ASYNC_OVERLAP:-0}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --check 0 --no-hbm-leg --sustain-seconds 0 --host-steps 0 "$@" > /dev/null 2> $OUT/p$i.err || echo "pass $i failed" >> $OUT/failed.txt
done
for K in $(echo "${KERNEL:-scan_kernel}" | tr ',' ' '); do   # KERNEL: one kernel or a comma-separated list (a summary each)
python - "$OUT" "$K" <<'PY'
import collections, csv, glob, json, sys
out, kernel = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r:
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {k: {"dispatches": len(v), "mean": sum(v) / len(v), "max": max(v)} for k, v in agg.items()}
res["_kernel"] = kernel
res["_scan_kernel_mean_ns_under_pmc"] = sum(dur) / max(1, len(dur))
json.dump(res, open(out + ("/sq_summary.json" if kernel == "scan_kernel" else "/sq_summary_%s.json" % kernel), "w"), indent=1)
print("==", kernel)
for k, v in sorted(res.items()):
    print(k, v if not isinstance(v, dict) else round(v["mean"]))
PY
done
rm -rf $OUT/p?/
