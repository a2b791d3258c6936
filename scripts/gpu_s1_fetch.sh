#!/bin/bash
# HBM bytes fetched by one strand1_kernel launch (125 M codes): rocprofv3 --pmc FETCH_SIZE (x2: the gfx950 correction for wide reads)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/s1_fetch; rm -rf $OUT; mkdir -p $OUT
python scripts/dev_strand1.py --codes ${CODES:-125000000} --calls 2 > $OUT/warm.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p -- python scripts/dev_strand1.py --codes ${CODES:-125000000} --calls 4 > $OUT/p.out 2> $OUT/p.err
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
vals = {}
for f in glob.glob(out + "/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        vals.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print("%-42s dispatches %4d  FETCH_SIZE mean %.1f KB -> x2 = %.1f MB per launch" % (k, len(v), sum(v) / len(v), 2 * sum(v) / len(v) / 1024))
PY
rm -rf $OUT/p
