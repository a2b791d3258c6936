#!/bin/bash
# strand1_kernel timings for the in-tree library and every variants/lib_*.so (125 M codes unless CODES is set)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/s1_variants.txt
for lib in "" variants/lib_*.so; do
  [ -z "$lib" ] || [ -e "$lib" ] || continue
  for codes in ${CODES:-125000000}; do
    DPQ_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 500 python scripts/dev_strand1.py --codes $codes --check 1 --tag "${lib:-in-tree}" 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/s1_variants.txt
  done
done
