#!/bin/bash
# Vector-memory side counters of the one-query strand pass (TA / TD / TCP / address translation), one rocprofv3 --pmc pass per group.
#   gpurun -- 'bash scripts/gpu_s1_mem_pmc.sh TAG [CODES]'
TAG=${1:-s1_mem}; CODES=${2:-125000000}; KERNEL=strand1_kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
python scripts/dev_strand1.py --codes $CODES --calls 2 > $OUT/warm.txt 2>&1   # fills the payload cache
i=0
# (at most two counters of a block per pass: more "exceeds the capabilities of the hardware to collect")
for grp in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python scripts/dev_strand1.py --codes $CODES --calls 6 > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i failed" >> $OUT/failed.txt; grep -m2 "error code\|exceeds" $OUT/p$i.err; break; }
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
json.dump(res, open(out + "/mem_summary.json", "w"), indent=1)
for k, v in sorted(res.items()):
    print(k, v if not isinstance(v, dict) else round(v["mean"]))
PY
[ -e $OUT/failed.txt ] && cat $OUT/failed.txt $OUT/p*.err | tail -20
rm -rf $OUT/p?/
