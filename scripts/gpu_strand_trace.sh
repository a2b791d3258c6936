#!/bin/bash
export DPQ_DEV=1
# per-launch durations of the stream pass's kernels (125 M codes, one query): the three filter levels separately
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/strand_trace
rm -rf $OUT && mkdir -p $OUT
Q=${Q:-1}
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python bench.py --codes ${CODES:-125000000} --data stream --queries $Q --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --check 0 --sustain-seconds 0 --host-steps 0 > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; }
python - <<PY
import csv, glob
rows=[]
for f in glob.glob("$OUT/tr/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "dpq::" in n: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.replace("void dpq::","").split("(")[0][:40]))
rows.sort()
for s,e,n in rows[-14:]:
    print("%-42s %8.1f us  (gap before next start: see order)" % (n, (e-s)/1e3))
print("span of the last 12 kernels: %.1f us" % ((rows[-1][1]-rows[-12][0])/1e3))
PY
rm -rf $OUT/tr
