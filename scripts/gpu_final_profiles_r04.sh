#!/bin/bash
# Round-4 closing evidence, regenerated from HEAD: every profiles/r04_* file a number in DESIGN.md refers to.
#   gpurun -- 'bash scripts/gpu_final_profiles_r04.sh a'   default workload: bench line, kernel stats + FETCH/WRITE, SQ counters
#   gpurun -- 'bash scripts/gpu_final_profiles_r04.sh b'   one query per call on 125 M codes (strand1_kernel): stats + FETCH/WRITE, SQ counters
#   gpurun -- 'bash scripts/gpu_final_profiles_r04.sh c'   M = 16 top-1000 (index built once, unprofiled) and M = 8 top-1000 lines with stats
#   gpurun -- 'bash scripts/gpu_final_profiles_r04.sh d'   full GPU suite + fuzz logs
# Results land in gpurun_out/r04final/ ; copy what is wanted into profiles/ (scripts/copy_r04_profiles.sh).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04final; mkdir -p $O
step() { echo "$(date +%T) $1" | tee -a $O/progress.txt; }
case "${1:-a}" in
a)
  step "default bench line"; timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/r04_bench_line_default.json 2> $O/bench_default.err; echo "rc=$?"
  step "default stats + pmc"; bash scripts/collect_pmc.sh r04_default > $O/pmc_default.log 2>&1; echo "rc=$?"
  step "default sq"; bash scripts/collect_sq_pmc.sh r04_scan_sq > $O/sq_default.log 2>&1; echo "rc=$?" ;;
b)
  step "125M one query: line"; timeout -k 10 600 python bench.py --codes 125000000 --data stream --queries 1 --steps 10 --warmup 2 --reps 3 --check 4 --min-check 4 --no-cpu-baseline --sustain-seconds 0 --host-steps 0 --no-hbm-leg > $O/r04_125M_stream_bench_line.json 2> $O/b125.err; echo "rc=$?"
  step "125M one query: stats + pmc"; bash scripts/collect_pmc.sh r04_125M_stream --codes 125000000 --data stream --queries 1 --check 0 > $O/pmc_125M.log 2>&1; echo "rc=$?"
  step "125M one query: sq"; bash scripts/gpu_s1_pmc.sh r04_sq_strand1 125000000 0 strand1_kernel > $O/sq_125M.log 2>&1; echo "rc=$?" ;;
c)
  step "m16 top-1000 line"; timeout -k 10 900 python bench.py --m 16 --topk 1000 --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg --index-dir /tmp/dpq_index_cache > $O/r04_m16_top1000_bench_line.json 2> $O/m16.err; echo "rc=$?"
  step "m16 top-1000 stats + pmc"; bash scripts/collect_pmc.sh r04_m16_top1000 --m 16 --topk 1000 > $O/pmc_m16.log 2>&1; echo "rc=$?"
  for k in 1000 2048; do step "top-$k line"; timeout -k 10 600 python bench.py --topk $k --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/r04_top${k}_bench_line.json 2> $O/top$k.err; echo "rc=$?"; done ;;
d)
  step "pytest"; timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r04_pytest_gpu.log 2>&1; tail -2 $O/r04_pytest_gpu.log
  step "fuzz"; timeout -k 10 400 python scripts/fuzz_parity.py 240 41 > $O/r04_fuzz.log 2>&1; tail -1 $O/r04_fuzz.log
  step "fuzz big"; DPQ_FUZZ_BIG=1 timeout -k 10 400 python scripts/fuzz_parity.py 240 42 > $O/r04_fuzz_big.log 2>&1; tail -1 $O/r04_fuzz_big.log ;;
esac
step "done ${1:-a}"
du -sh gpurun_out | tail -1
