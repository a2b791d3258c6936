#!/bin/bash
# timing experiments (wrong results by design): strand1_kernel with parts left out, 125 M codes, no exact checks
cd "$GRAFT_REPO_ROOT" || exit 1
export DPQ_DEV=1 DPQ_S1_DEBUG=1
mkdir -p gpurun_out; : > gpurun_out/s1_skip.txt
for lib in "" variants/lib_*.so; do
  [ -z "$lib" ] || [ -e "$lib" ] || continue
  DPQ_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 300 python scripts/dev_strand1.py --codes 125000000 --tag "${lib:-in-tree}" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200 | tee -a gpurun_out/s1_skip.txt
done
