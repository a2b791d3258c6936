#!/bin/bash
# (needs scripts/experiments/r03_checker_wavefronts.patch applied: the switches and variants it builds live there)
# what an exact check costs when its gathers diverge over the queries' tables (x1) and when all lanes read ONE query's table (x2)
mkdir -p gpurun_out
: > gpurun_out/extra_check.txt
for cfg in "M=8 K=100" "M=8 K=1000" "M=16 K=1000"; do
  for lib in "" variants/lib_x1.so variants/lib_x2.so; do
    echo "== $cfg ${lib:-in-tree}" | tee -a gpurun_out/extra_check.txt
    env $cfg ${lib:+DPQ_LIB_PATH=$PWD/$lib} timeout -k 10 200 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/extra_check.txt
  done
done
