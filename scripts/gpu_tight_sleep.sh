#!/bin/bash
# helper wavefront: pause between two looks (s_sleep units of 64 cycles; 32 in-tree) with one-unit tightening steps
mkdir -p gpurun_out
: > gpurun_out/tight_sleep.txt
for cfg in "M=8 K=100" "M=8 K=1000"; do
  for lib in "" variants/lib_ts16.so variants/lib_ts64.so; do
    echo "== $cfg ${lib:-ts32}" | tee -a gpurun_out/tight_sleep.txt
    env $cfg ${lib:+DPQ_LIB_PATH=$PWD/$lib} timeout -k 10 100 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | cut -c30-330 | tee -a gpurun_out/tight_sleep.txt
  done
done
