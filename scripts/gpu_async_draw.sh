#!/bin/bash
# segment draw one call ahead + segment numbers loaded lazily (in-tree) against the draw on demand (variants/lib_sync.so)
# (needs scripts/experiments/r03_async_segment_draw.patch applied; builds variants/lib_sync.so with -DDPQ_ASYNC_DRAW=0)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not 125m and not config4 and not config3 and not 12p5" > gpurun_out/ad_pytest.txt 2>&1 || { tail -30 gpurun_out/ad_pytest.txt; exit 1; }
tail -2 gpurun_out/ad_pytest.txt
: > gpurun_out/async_draw.txt
for cfg in "M=8 K=100" "M=8 K=10" "M=8 K=1000" "M=16 K=1000" "M=8 K=100"; do
  for lib in "" variants/lib_sync.so; do
    echo "== $cfg ${lib:-async}" | tee -a gpurun_out/async_draw.txt
    env $cfg ${lib:+DPQ_LIB_PATH=$PWD/$lib} timeout -k 10 100 python scripts/dev_scan_variants.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a gpurun_out/async_draw.txt
  done
done
