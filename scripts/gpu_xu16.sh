#!/bin/bash
# M = 16: filter units per tightening step (XU; 7 steps at most) at top-1000 and top-100
mkdir -p gpurun_out
: > gpurun_out/xu16.txt
for cfg in "M=16 K=1000" "M=16 K=100"; do
  for lib in "" variants/lib_xu4.so variants/lib_xu6.so; do
    echo "== $cfg ${lib:-xu8}" | tee -a gpurun_out/xu16.txt
    env $cfg ${lib:+DPQ_LIB_PATH=$PWD/$lib} timeout -k 10 100 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | cut -c30-330 | tee -a gpurun_out/xu16.txt
  done
done
