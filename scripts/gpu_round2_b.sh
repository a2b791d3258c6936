#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02b_pytest_gpu.log
tail -25 gpurun_out/r02b_pytest_gpu.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; echo "bench rc=$?"
cat gpurun_out/r02b_bench.json | cut -c1-300; tail -3 gpurun_out/r02b_bench.err
TAG=r02b timeout -k 10 300 python scripts/dev_limiter.py > gpurun_out/r02b_limiter.log 2>&1; echo "limiter rc=$?"
grep -v "^stamps\|amdgpu.ids" gpurun_out/r02b_limiter.log | cut -c1-600
