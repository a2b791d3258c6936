#!/bin/bash
# (needs scripts/experiments/r03_checker_wavefronts.patch applied: the switches and variants it builds live there)
mkdir -p gpurun_out
export DPQ_DEV=1
: > gpurun_out/chk_stats.txt
for cfg in "M=8 K=100 DPQ_CHECKERS=4" "M=8 K=1000 DPQ_CHECKERS=6" "M=16 K=1000 DPQ_CHECKERS=4"; do
  echo "== $cfg" | tee -a gpurun_out/chk_stats.txt
  env $cfg DPQ_LIB_PATH=$PWD/variants/lib_st.so timeout -k 5 100 python scripts/dev_scan_variants.py 2>&1 | grep CHKSTAT | sed -n '5,6p' | tee -a gpurun_out/chk_stats.txt
  for lib in "" variants/lib_pp3.so; do
  env $cfg ${lib:+DPQ_LIB_PATH=$PWD/$lib} timeout -k 5 100 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a gpurun_out/chk_stats.txt
  done
done
