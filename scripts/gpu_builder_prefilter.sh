#!/bin/bash
# round 4: the GPU builder's per-subset hash pre-filter (find_edges_gpu): identical trees (builder tests), and the 1 M x M = 16
# index build under rocprofv3 --kernel-trace --stats (dispatches, GPU time), with the pre-filter and (PF0=1) without
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/bpf; mkdir -p $O
step() { echo "$(date +%T) $1" | tee -a $O/progress.txt; }
step "builder tests"
timeout -k 10 600 python -m pytest tests/test_builder.py tests/test_hand_derived_builder.py -q -x -m gpu > $O/pytest_builder.log 2>&1 || { tail -30 $O/pytest_builder.log; exit 1; }
tail -2 $O/pytest_builder.log
for pf in ${PFS:-1}; do
  step "1 M x M = 16 build, pre-filter $pf"
  rm -rf /tmp/idx16_$pf
  ( export DPQ_DEV=1 DPQ_BUILD_PREFILTER=$pf; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$pf -- python bench.py --m 16 --topk 1000 --index-dir /tmp/idx16_$pf --build-only > $O/build$pf.json 2> $O/build$pf.err ) ; echo "rc=$?"
  cat $O/build$pf.json | tail -1
  f=$(ls $O/prof$pf/*/*kernel_stats.csv | head -1)
  python - "$f" <<'PY' | tee $O/stats$pf.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = sum(int(r["Calls"]) for r in rows); ns = sum(float(r["TotalDurationNs"]) for r in rows)
print("dispatches %d, GPU time %.2f s" % (calls, ns / 1e9))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("  %-70s %8d calls %8.3f s" % (r["Name"][:70], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e9))
PY
  cp "$f" $O/kernel_stats_pf$pf.csv; rm -rf $O/prof$pf
done
step done
