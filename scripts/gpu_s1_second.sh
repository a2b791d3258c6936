#!/bin/bash
# strand1_kernel: parity tests that reach it, then timings (in-tree + variants) on 32 M and 125 M codes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_query or stream_pass_on_a_prefix" > gpurun_out/s1_pytest.txt 2>&1
rc=$?
tail -5 gpurun_out/s1_pytest.txt
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/s1_times.txt
for lib in "" variants/lib_*.so; do
  [ -z "$lib" ] || [ -e "$lib" ] || continue
  for codes in 32000000 125000000; do
    DPQ_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 500 python scripts/dev_strand1.py --codes $codes --check 1 --tag "${lib:-in-tree}" 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/s1_times.txt
  done
done
