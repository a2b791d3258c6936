#!/bin/bash
# first GPU call of round 2: parity tests of the rewritten scan path, limiter probe, bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1 || true
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a_pytest_gpu.log
tail -5 gpurun_out/r02a_pytest_gpu.log
TAG=r02a timeout -k 10 300 python scripts/dev_limiter.py > gpurun_out/r02a_limiter.log 2>&1; echo "limiter rc=$?"
tail -12 gpurun_out/r02a_limiter.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench rc=$?"
cat gpurun_out/r02a_bench.json | cut -c1-1500
