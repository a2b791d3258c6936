#!/bin/bash
# survivor bits a lane checks per refine round (RB = 2 in-tree; variants rb3 / rb4) at small and large top_k
mkdir -p gpurun_out
: > gpurun_out/refine_bits.txt
for cfg in "M=8 K=1000" "M=16 K=1000" "M=8 K=100"; do
  for lib in "" variants/lib_rb3.so variants/lib_rb4.so; do
    echo "== $cfg ${lib:-rb2}" | tee -a gpurun_out/refine_bits.txt
    env $cfg ${lib:+DPQ_LIB_PATH=$PWD/$lib} timeout -k 10 100 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | cut -c30-330 | tee -a gpurun_out/refine_bits.txt
  done
done
