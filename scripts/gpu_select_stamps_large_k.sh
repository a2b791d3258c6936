#!/bin/bash
# per-block phase stamps of the select kernel's last level over top_k, radix select + bitonic sort against the bucket sort
# (a block that takes the exact way at top_k > 256 leaves no end stamp: its lifetime prints as ~1.8e17)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/selk; mkdir -p $O
for cfg in ${CFGS:-8:100 8:512 8:1000 8:2048 16:1000}; do
  IFS=: read m k <<< "$cfg"
  echo "=== M=$m top-$k" | tee -a $O/stamps.txt
  M=$m K=$k timeout -k 10 400 python scripts/dev_boot_stamps.py 1:::0 1:::1 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
done
