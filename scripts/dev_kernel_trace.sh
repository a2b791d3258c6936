#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): per-launch durations of the query-path kernels of the last bench steps.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/kt
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python bench.py --steps 3 --warmup 1 --check 0 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/kt.err
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "dpq::s" in r["Kernel_Name"] or "lut_build" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-8:]:
    print("%-28s %7.1f us" % (r["Kernel_Name"].split("(")[0][-28:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0))
PY
rm -rf gpurun_out/kt
