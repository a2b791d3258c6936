#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): SQ / LDS counters of the scan kernel on the default workload,
# one rocprofv3 --pmc pass per counter group (only --kernel-trace beside it, as the pool requires).
#   gpurun -- 'bash scripts/collect_sq_pmc.sh r01_sq'
set -o pipefail
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0 > /dev/null 2> $OUT/p$i.err || echo "pass $i failed" >> $OUT/failed.txt
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "scan_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {"dispatches": len(v), "max": max(v), "sum": sum(v), "values_last3": v[-3:]} for k, v in agg.items()}
json.dump(res, open(out + "/sq_summary.json", "w"), indent=1)
for k, v in sorted(res.items()):
    print(k, v["values_last3"])
PY
