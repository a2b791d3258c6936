#!/bin/bash
# M = 8: filter units per tightening step (2 in-tree, variant 1) at top-100 / top-1000 / top-10
mkdir -p gpurun_out
: > gpurun_out/xu8.txt
for cfg in "M=8 K=100" "M=8 K=1000" "M=8 K=10"; do
  for lib in "" variants/lib_x1s23.so; do
    echo "== $cfg ${lib:-xu2}" | tee -a gpurun_out/xu8.txt
    env $cfg ${lib:+DPQ_LIB_PATH=$PWD/$lib} timeout -k 10 100 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | cut -c30-330 | tee -a gpurun_out/xu8.txt
  done
done
