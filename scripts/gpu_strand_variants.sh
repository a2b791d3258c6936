#!/bin/bash
# big-level time of the strand pass (125 M codes, one query) for the in-tree library and every variants/lib_*.so
for lib in "" variants/lib_*.so; do
  [ -z "$lib" ] || [ -e "$lib" ] || continue
  echo "== ${lib:-in-tree}"
  DPQ_LIB_PATH=${lib:+$PWD/$lib} Q=${Q:-1} bash scripts/gpu_strand_trace.sh | grep "strand_kernel\|stream_kernel" | tail -3
done 2>&1 | tee gpurun_out/strand_variants.txt
