#!/bin/bash
# SQ / LDS counters of the one-query strand pass (scripts/dev_strand1.py workload), one rocprofv3 --pmc pass per group.
#   gpurun -- 'bash scripts/gpu_s1_pmc.sh TAG [CODES] [FLAGS] [KERNEL]'
TAG=${1:-s1_pmc}; CODES=${2:-32000000}; FLAGS=${3:-0}; KERNEL=${4:-strand1_kernel}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
[ -n "$LIST" ] && rocprofv3 -L > $OUT/counters_list.txt 2>&1
python scripts/dev_strand1.py --codes $CODES --calls 2 --flags $FLAGS > $OUT/warm.txt 2>&1   # fills the payload cache
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM" \
           ${EXTRA:+"$EXTRA"}; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python scripts/dev_strand1.py --codes $CODES --calls 6 --flags $FLAGS > $OUT/p$i.out 2> $OUT/p$i.err || echo "pass $i failed" >> $OUT/failed.txt
done
python - "$OUT" "$KERNEL" <<'PY'
import collections, csv, glob, json, sys
out, kernel = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p*/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {k: {"dispatches": len(v), "mean": sum(v) / len(v), "max": max(v)} for k, v in agg.items()}
res["_kernel"] = kernel
res["_kernel_mean_ns_under_pmc"] = sum(dur) / max(1, len(dur))
json.dump(res, open(out + "/sq_summary.json", "w"), indent=1)
for k, v in sorted(res.items()):
    print(k, v if not isinstance(v, dict) else round(v["mean"]))
PY
rm -rf $OUT/p?/
