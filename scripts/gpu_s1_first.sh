#!/bin/bash
# first GPU contact of strand1_kernel: the parity tests that reach it, then timings on 32 M and 125 M codes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_query or stream_pass_on_a_prefix" > gpurun_out/s1_pytest.txt 2>&1
rc=$?
tail -15 gpurun_out/s1_pytest.txt
[ $rc -eq 0 ] || exit $rc
for flags in 0 128; do
  timeout -k 10 300 python scripts/dev_strand1.py --codes 32000000 --check 2 --flags $flags --tag in-tree 2>&1 | tail -4
done | tee gpurun_out/s1_times.txt
for flags in 0 128; do
  timeout -k 10 500 python scripts/dev_strand1.py --codes 125000000 --check 1 --flags $flags --tag in-tree 2>&1 | tail -3
done | tee -a gpurun_out/s1_times.txt
