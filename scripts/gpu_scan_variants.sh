#!/bin/bash
# every library under variants/ (and the in-tree one, with and without the in-scan tightening) through scripts/dev_scan_variants.py
mkdir -p gpurun_out
python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/scan_variants.txt
DPQ_DEV=1 DPQ_TIGHTEN=0 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | sed 's/^in-tree /in-tree-notighten/' | tee -a gpurun_out/scan_variants.txt
for lib in variants/lib_*.so; do
  [ -e "$lib" ] || continue
  DPQ_LIB_PATH=$PWD/$lib timeout -k 10 120 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/scan_variants.txt
done
