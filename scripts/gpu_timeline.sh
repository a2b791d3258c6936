#!/bin/bash
# kernel timeline (start / end per dispatch) of the last pipelined steps of the bench: where the gaps are
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/timeline
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python bench.py --steps 12 --warmup 3 --reps 2 --no-cpu-baseline --check 0 --sustain-seconds 0 --host-steps 0 $BENCH_ARGS > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "dpq::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the second timed repetition: the last 12 steps before the synchronous aux pass; find by taking a window near the end
names = [r["Kernel_Name"].replace("void dpq::", "").replace("dpq::", "").split("(")[0].split("<")[0] for r in rows]
scans = [i for i, n in enumerate(names) if n == "scan_kernel"]
# the aux pass has 8 scans at the end (synchronous); take the 10 scans before them
sel = scans[-18:-8]
i0, i1 = sel[0] - 6, sel[-1] + 3
t0 = int(rows[i0]["Start_Timestamp"])
out = []
for i in range(i0, i1):
    r = rows[i]
    out.append("%9.1f %9.1f  %6.1f us  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), names[i]))
open("$OUT/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:70]))
PY
rm -rf $OUT/trace
