#!/bin/bash
# Developer A/B: builds libdeltapq_amd.so variants with -D switches into variants/ (git-ignored; they travel with gpurun).
#   bash scripts/build_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"-DBAZ=0" ...
cd "$(dirname "$0")/../deltapq_amd/csrc" || exit 1
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value $flags \
      -no-hip-rt -shared -o ../../variants/lib_$name.so dpq_kernels.hip dpq_build_gpu.hip dpq_capi.cpp dpq_format.cpp dpq_build.cpp 2>&1 | grep -E "error" ; echo "built $name ($flags)" ) &
done
wait
